// blokus.hip -- the Blokus environment (rules: az_blokus_rules.h) as lockstep kernels over G concurrent
// games, one game per 64-lane wavefront, and its C ABI (az_blokus_*).  The Env members this stands for are
// reset / step / possible_actions / get_board / get_player (call sites mcts.hpp:251-271,427;
// generate.cpp:75,103); the reference's own Blokus source is absent, so the rules are this build's.
//
// State in HBM, structure-of-arrays over the games: per player 20 row bit boards (u32, bit x = column x),
// the pieces-played masks, and -- kept current for the player to move -- the legal-action set as row masks
// [91 orientations][20 rows] (bit x of word (o, y) = action o*400 + y*20 + x) with its count.
// Legality for player q comes from two derived boards,
//   free[r]    = cells that are empty and not edge-adjacent to q's squares
//   contact[r] = cells diagonal to q's squares (q's start corner before q's first piece)
// and is evaluated for a whole board row at a time: for orientation o anchored in row y, with cells (cy, cx),
//   legal_x(o, y) = AND_cells (free[y + cy] >> cx)  &  OR_cells (contact[y + cy] >> cx)
// (a shifted-in zero is an off-board cell), so one lane settles 20 placements in about eight operations per
// piece cell; the 91 x 20 (orientation, row) pairs are strided over the 64 lanes, the counts come from popcounts.
#include "../../include/az_hip.h"
#include "az_blokus_rules.h"
#include "az_dev.h"

#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

extern "C" int az_internal_fail(int code, const char* msg);

#define AZB_WORDS AZB_N                                // one u32 row mask per board row of an orientation
#define AZB_SET_WORDS (AZB_ORIENTS * AZB_WORDS)

__constant__ AzbOrient c_orient[AZB_ORIENTS];
__constant__ uint8_t c_piece_size[AZB_PIECES];

struct AzbDev {
    int G;
    unsigned long long seed;
    int slot_base;
    uint32_t* own;          // [G][4][20]
    uint32_t* used;         // [G][4]
    uint8_t*  last_piece;   // [G][4]
    uint8_t*  to_move;      // [G]
    uint8_t*  over;         // [G]
    int*      moves;        // [G] pieces placed so far
    uint16_t* hist;         // [G][84] the actions played since reset: Env::get_id (mcts.hpp:388)
    int*      legal_count;  // [G] legal actions of the player to move (0 once over)
    uint32_t* legal_bits;   // [G][91][20] row masks
    float*    final_reward; // [G][4]
    // ---- search (MCTS) state, allocated by az_blokus_search_config ----
    float vl; int c_puct_i;
    int edge_cap;           // arena slots per game
    int use_noise;
    int L;                  // descents per game and round (az_blokus_search_leaves): evaluation slot g * L + k
    AzEdge*   edges;        // [G][edge_cap]  a node = one header slot {n = child count, child = player to move} + its child edges
    uint16_t* eact;         // [G][edge_cap]  action of each child edge
    // tree reuse across searches (__make_root + find, mcts.hpp:385-421), az_blokus_search_reuse:
    int reuse;              // 1: a search begins by looking the position up in the kept tree
    AzEdge*   edges1;       // [G][edge_cap]  the second arena: a kept subtree is copied from one to the other, its root to slot 0
    uint16_t* eact1;
    int*      arena;        // [G] which arena holds the game's tree
    int*      root_moves;   // [G] pieces on the board at the tree's root, -1: no tree
    uint16_t* root_hist;    // [G][84] the root's move history
    int*      queue;        // [G][q_cap] header slots of the subtree being copied, in BFS order
    int q_cap;
    uint8_t*  warm;         // [G] self-play: the game's next search is its warm-up (generate.cpp:77), no move follows it
    int*      n_slots;      // [G] arena slots in use
    int*      n_nodes;      // [G] expanded nodes
    int*      root_n;       // [G]
    int*      t_status;     // [G] 1 while the search runs
    int*      sims_done;    // [G]
    int*      sims_target;  // [G]
    // per evaluation slot (S = G * L of them):
    int*      pending;      // [S] AZ_PEND_*
    int*      path_len;     // [S]
    int*      path;         // [S][AZB_PATH_CAP] edge slots root -> leaf
    uint8_t*  path_owner;   // [S][AZB_PATH_CAP] who chose at each level
    float*    pend_result;  // [S][4] reward of a terminal leaf
    uint32_t* leaf_bits;    // [S][91][20] legal set of the leaf awaiting its evaluation
    int*      leaf_count;   // [S]
    uint8_t*  leaf_player;  // [S]
    uint8_t*  leaf_planes;  // [S][5][400] evaluator input
    float*    policy;       // [S][91*400] evaluator output
    float*    value;        // [S][4]
    float*    noise;        // [G][AZB_NOISE_CAP]
    int*      restarts;     // [G] searches restarted because the arena was full
    int*      search_restarts;  // [G] ... in the current search
    int*      truncated;    // [G] searches ended early because the arena filled up a second time
    unsigned* noise_ctr;    // [G] device-noise draws so far (use_noise == 2)
    // ---- self-play state, allocated by az_blokus_selfplay_config ----
    int sp_k;               // policy entries kept per row (a root child needs a visit to have one): max_budget, 4 x with reuse
    int ring_cap;           // rows of the replay ring
    float*     sp_avg_len;      // [G] running mean game length (generate.cpp:25-29,69)
    long long* sp_game_count;   // [G]
    unsigned*  sp_sample_ctr;   // [G]
    uint8_t*   row_planes;      // [G][84][5*400] rows of the game in progress (buffer.temporary_append, generate.cpp:85)
    uint8_t*   row_tomove;      // [G][84]
    int*       row_cnt;         // [G][84] policy entries of the row
    uint16_t*  row_act;         // [G][84][sp_k] their actions, ascending
    float*     row_p;           // [G][84][sp_k] n_child / n_root
    uint8_t*   ring_planes;     // [ring_cap][5*400] rows of finished games (buffer.send_reward, generate.cpp:110)
    uint8_t*   ring_tomove;
    int*       ring_cnt;
    uint16_t*  ring_act;
    float*     ring_p;
    float*     ring_reward;     // [ring_cap][4] by absolute player
    unsigned long long* sp_counters;    // 0 rows in the ring, 1 rows dropped (ring full), 2 moves played, 3 games finished,
                                        // 4 moves NOT played because the game's root did not fit its arena,
                                        // 5 rows whose policy had more visited children than a row holds (sp_k)
};
#define AZB_MAX_MOVES 84                                // 4 players x 21 pieces
#define AZB_PATH_CAP 88                                 // a game has at most 84 pieces
#define AZB_NOISE_CAP 4096

// ---------------------------------------------------------------------------------------------- device
// The orientation table in LDS, 8 bytes each: x = piece | h << 8 | n << 16 | cell0 << 24, y = cells 1..4, a cell
// as cy << 4 | cx (lanes index it by different orientations: LDS serves that, the constant cache would not).
__device__ void azb_load_table(uint2* s_tab) {
    for (int o = threadIdx.x; o < AZB_ORIENTS; o += 64) {
        uint32_t c[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < c_orient[o].n; ++i) c[i] = (uint32_t)(c_orient[o].cy[i] << 4 | c_orient[o].cx[i]);
        s_tab[o] = make_uint2(c_orient[o].piece | (uint32_t)c_orient[o].h << 8 | (uint32_t)c_orient[o].n << 16 | c[0] << 24,
                              c[1] | c[2] << 8 | c[3] << 16 | c[4] << 24);
    }
    __syncthreads();
}

// per-orientation counts of a legal set
__device__ void azb_count(const uint32_t* s_bits, int* s_cnt) {
    for (int o = threadIdx.x; o < AZB_ORIENTS; o += 64) {
        int c = 0;
        for (int y = 0; y < AZB_N; ++y) c += __popc(s_bits[o * AZB_WORDS + y]);
        s_cnt[o] = c;
    }
    __syncthreads();
}

// Legal-action set of player q on the position in s_own: fills s_bits [91][20] and s_cnt [91]; returns the
// total.  All 64 lanes take part.
__device__ int azb_scan(const uint32_t* s_own, uint32_t used_q, int q, uint2* s_rows, const uint2* s_tab,
                        uint32_t* s_bits, int* s_cnt) {
    const int lane = threadIdx.x;
    const uint32_t full = (1u << AZB_N) - 1u;
    if (lane < AZB_N) {
        const int r = lane;
        const uint32_t mine = s_own[q * AZB_N + r];
        const uint32_t up = r > 0 ? s_own[q * AZB_N + r - 1] : 0u, dn = r < AZB_N - 1 ? s_own[q * AZB_N + r + 1] : 0u;
        const uint32_t occ = s_own[r] | s_own[AZB_N + r] | s_own[2 * AZB_N + r] | s_own[3 * AZB_N + r];
        const uint32_t forb = occ | (mine << 1) | (mine >> 1) | up | dn;
        uint32_t c = ((up << 1) | (up >> 1) | (dn << 1) | (dn >> 1)) & full;
        if (used_q == 0u) {                                          // first piece: the player's own corner
            const int cy = q < 2 ? 0 : AZB_N - 1, cx = (q == 0 || q == 3) ? 0 : AZB_N - 1;
            c = r == cy ? 1u << cx : 0u;
        }
        s_rows[r] = make_uint2(~forb & full, c);
    }
    __syncthreads();
    int total = 0;
    for (int i = lane; i < AZB_SET_WORDS; i += 64) {
        const int o = i / AZB_N, y = i - AZB_N * o;
        const uint2 tab = s_tab[o];
        const int piece = tab.x & 0xFF, h = tab.x >> 8 & 0xFF, n = tab.x >> 16 & 0xFF;
        uint32_t mask = 0u;
        if (!(used_q >> piece & 1u) && y + h <= AZB_N) {
            const unsigned long long cells = (unsigned long long)(tab.x >> 24) | (unsigned long long)tab.y << 8;
            uint32_t fit = full, touch = 0u;
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (k < n) {
                    const int c = (int)(cells >> (8 * k)) & 0xFF;
                    const uint2 rw = s_rows[y + (c >> 4)];
                    fit &= rw.x >> (c & 15);
                    touch |= rw.y >> (c & 15);
                }
            mask = fit & touch;
        }
        s_bits[i] = mask;
        total += __popc(mask);
    }
    for (int o2 = 32; o2 > 0; o2 >>= 1) total += __shfl_xor(total, o2, 64);
    __syncthreads();
    azb_count(s_bits, s_cnt);
    return total;
}

// The k-th legal action in index order (k < total), from s_bits / s_cnt, by the whole wave: an inclusive scan
// over the orientation counts (two trips of 64) finds the orientation, a scan over its 20 row popcounts the
// row, a few bit clears the column.  Every lane returns the action.
__device__ int azb_kth_action(const uint32_t* s_bits, const int* s_cnt, int k) {
    const int lane = threadIdx.x;
    int o = -1;
    for (int base = 0; base < AZB_ORIENTS && o < 0; base += 64) {
        int x = base + lane < AZB_ORIENTS ? s_cnt[base + lane] : 0;
        const int own = x;
        for (int d = 1; d < 64; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
        const unsigned long long hit = __ballot(x > k);
        if (hit) {
            const int l = __ffsll((long long)hit) - 1;
            o = base + l;
            k -= __shfl(x - own, l, 64);                     // actions before orientation o
        } else k -= __shfl(x, 63, 64);
    }
    int x = lane < AZB_N ? __popc(s_bits[o * AZB_WORDS + lane]) : 0;
    const int own = x;
    for (int d = 1; d < 32; d <<= 1) { const int y = __shfl_up(x, d, 64); if (lane >= d) x += y; }
    const unsigned long long hit = __ballot(lane < AZB_N && x > k);
    const int y = __ffsll((long long)hit) - 1;
    k -= __shfl(x - own, y, 64);
    uint32_t v = s_bits[o * AZB_WORDS + y];
    for (; k > 0; --k) v &= v - 1u;
    return o * AZB_CELLS + AZB_N * y + (__ffs((int)v) - 1);
}

// Player p's piece `action` onto the position in LDS.
__device__ void azb_place(uint32_t* s_own, uint32_t* s_used, int* s_last, int p, int action) {
    const int lane = threadIdx.x;
    const int o = action / AZB_CELLS, y0 = action % AZB_CELLS / AZB_N, x0 = action % AZB_N;
    __syncthreads();
    if (lane < c_orient[o].h) s_own[p * AZB_N + y0 + lane] |= (uint32_t)c_orient[o].rows[lane] << x0;
    if (lane == 0) { s_used[p] |= 1u << c_orient[o].piece; s_last[p] = c_orient[o].piece; }
    __syncthreads();
}

// After player p moved: the next player (cyclically) who has a legal action, with that action set in
// s_bits / s_cnt and its size in *cnt; -1 when nobody can move.
__device__ int azb_next_player(const uint32_t* s_own, const uint32_t* s_used, int p, uint2* s_rows, const uint2* s_tab,
                               uint32_t* s_bits, int* s_cnt, int* cnt) {
    for (int k = 1; k <= AZB_PLAYERS; ++k) {
        const int q = (p + k) & 3;
        *cnt = azb_scan(s_own, s_used[q], q, s_rows, s_tab, s_bits, s_cnt);
        if (*cnt > 0) return q;
    }
    *cnt = 0;
    return -1;
}

// End of the game: the reward of every player from the hands (every lane computes the same four values).
__device__ void azb_final_reward_dev(const uint32_t* s_used, const int* s_last, float r[4]) {
    int score[4], best = -1000;
    bool all_equal = true;
    for (int q = 0; q < 4; ++q) {
        int left = 0;
        for (int pc = 0; pc < AZB_PIECES; ++pc) if (!(s_used[q] >> pc & 1u)) left += c_piece_size[pc];
        score[q] = left ? -left : 15 + (s_last[q] == 0 ? 5 : 0);
        best = max(best, score[q]);
        all_equal = all_equal && score[q] == score[0];
    }
    for (int q = 0; q < 4; ++q) r[q] = all_equal ? 0.f : (score[q] == best ? 1.f : -1.f);
}

// Places `action` (known legal) for the player to move, then hands the turn to the next player who can move,
// or ends the game.  s_own / s_used / s_last / s_meta {to_move, over, pieces placed, legal count} hold the
// game (and stay current, so a kernel can play on without re-reading HBM); the state arrays are updated too.
__device__ void azb_play(const AzbDev& d, int g, int action, uint32_t* s_own, uint2* s_rows, const uint2* s_tab,
                         uint32_t* s_bits, int* s_cnt, uint32_t* s_used, int* s_last, int* s_meta, float* reward, uint8_t* done) {
    const int lane = threadIdx.x;
    const int p = s_meta[0];
    azb_place(s_own, s_used, s_last, p, action);
    int cnt = 0;
    const int next = azb_next_player(s_own, s_used, p, s_rows, s_tab, s_bits, s_cnt, &cnt);
    // write the position back: only player p's rows and masks changed
    if (lane < AZB_N) d.own[((size_t)g * 4 + p) * AZB_N + lane] = s_own[p * AZB_N + lane];
    for (int i = lane; i < AZB_SET_WORDS; i += 64) d.legal_bits[(size_t)g * AZB_SET_WORDS + i] = next >= 0 ? s_bits[i] : 0u;
    float r[4] = {0.f, 0.f, 0.f, 0.f};
    if (next < 0) azb_final_reward_dev(s_used, s_last, r);
    __syncthreads();
    if (lane == 0) {
        d.used[g * 4 + p] = s_used[p];
        d.last_piece[g * 4 + p] = (uint8_t)s_last[p];
        d.hist[(size_t)g * AZB_MAX_MOVES + s_meta[2]] = (uint16_t)action;
        s_meta[2] += 1; d.moves[g] = s_meta[2];
        s_meta[3] = cnt; d.legal_count[g] = cnt;
        if (next >= 0) { s_meta[0] = next; d.to_move[g] = (uint8_t)next; }
        else {                                                       // nobody can move: the hands are scored
            s_meta[1] = 1; d.over[g] = 1;
            for (int q = 0; q < 4; ++q) d.final_reward[g * 4 + q] = r[q];
        }
        if (reward) for (int q = 0; q < 4; ++q) reward[g * 4 + q] = r[q];
        if (done) done[g] = next < 0;
    }
    __syncthreads();
}

#define AZB_SMEM_DECL \
    __shared__ uint32_t s_own[4 * AZB_N]; __shared__ uint2 s_rows[AZB_N]; __shared__ uint2 s_tab[AZB_ORIENTS]; \
    __shared__ uint32_t s_bits[AZB_SET_WORDS]; __shared__ int s_cnt[AZB_ORIENTS]; __shared__ uint32_t s_used[4]; \
    __shared__ int s_last[4]; __shared__ int s_meta[4]

__device__ void azb_load(const AzbDev& d, int g, uint32_t* s_own, uint32_t* s_used, int* s_last, int* s_meta) {
    for (int i = threadIdx.x; i < 4 * AZB_N; i += 64) s_own[i] = d.own[(size_t)g * 4 * AZB_N + i];
    if (threadIdx.x < 4) { s_used[threadIdx.x] = d.used[g * 4 + threadIdx.x]; s_last[threadIdx.x] = d.last_piece[g * 4 + threadIdx.x]; }
    if (threadIdx.x == 0) { s_meta[0] = d.to_move[g]; s_meta[1] = d.over[g]; s_meta[2] = d.moves[g]; s_meta[3] = d.legal_count[g]; }
    __syncthreads();
}

// Env::reset for every game, including the first player's legal set.
__device__ void azb_reset_game(const AzbDev& d, int g, uint32_t* s_own, uint32_t* s_used, uint2* s_rows, const uint2* s_tab,
                               uint32_t* s_bits, int* s_cnt, bool keep_reward) {
    const int lane = threadIdx.x;
    __syncthreads();
    for (int i = lane; i < 4 * AZB_N; i += 64) { s_own[i] = 0u; d.own[(size_t)g * 4 * AZB_N + i] = 0u; }
    if (lane < 4) {
        s_used[lane] = 0u; d.used[g * 4 + lane] = 0u; d.last_piece[g * 4 + lane] = 0xFF;
        if (!keep_reward) d.final_reward[g * 4 + lane] = 0.f;
    }
    __syncthreads();
    const int cnt = azb_scan(s_own, 0u, 0, s_rows, s_tab, s_bits, s_cnt);
    for (int i = lane; i < AZB_SET_WORDS; i += 64) d.legal_bits[(size_t)g * AZB_SET_WORDS + i] = s_bits[i];
    if (lane == 0) { d.to_move[g] = 0; d.over[g] = 0; d.moves[g] = 0; d.legal_count[g] = cnt; }
    __syncthreads();
}

extern "C" __global__ void __launch_bounds__(64) azb_reset_kernel(AzbDev d) {
    AZB_SMEM_DECL;
    (void)s_last; (void)s_meta;
    azb_load_table(s_tab);
    azb_reset_game(d, blockIdx.x, s_own, s_used, s_rows, s_tab, s_bits, s_cnt, false);
}

// Env::step, batched: actions[g] = o*400 + y*20 + x, or -1 to leave game g alone.
extern "C" __global__ void __launch_bounds__(64)
azb_step_kernel(AzbDev d, const int* actions, float* reward, uint8_t* done, int* illegal) {
    AZB_SMEM_DECL;
    const int g = blockIdx.x, lane = threadIdx.x;
    const int action = actions[g];
    if (lane < 4) reward[g * 4 + lane] = 0.f;
    if (lane == 0) done[g] = d.over[g];
    if (action < 0) return;
    bool ok = action < AZB_ACTIONS && !d.over[g];
    if (ok) {
        const int o = action / AZB_CELLS, cell = action % AZB_CELLS;
        ok = d.legal_bits[(size_t)g * AZB_SET_WORDS + o * AZB_WORDS + cell / AZB_N] >> (cell % AZB_N) & 1u;
    }
    if (!ok) { if (lane == 0) atomicAdd(illegal, 1); return; }
    azb_load_table(s_tab);
    azb_load(d, g, s_own, s_used, s_last, s_meta);
    azb_play(d, g, action, s_own, s_rows, s_tab, s_bits, s_cnt, s_used, s_last, s_meta, reward, done);
}

// `steps` uniformly random legal moves per game, entirely on the device: the k-th legal action in index
// order, k = (u * count) >> 32 with u = word 0 of Philox(seed, stream = slot_base + g, counter = pieces placed).
extern "C" __global__ void __launch_bounds__(64) azb_playout_kernel(AzbDev d, int steps) {
    AZB_SMEM_DECL;
    __shared__ int s_pick;
    const int g = blockIdx.x, lane = threadIdx.x;
    if (d.over[g]) return;
    azb_load(d, g, s_own, s_used, s_last, s_meta);
    azb_load_table(s_tab);
    for (int i = lane; i < AZB_SET_WORDS; i += 64) s_bits[i] = d.legal_bits[(size_t)g * AZB_SET_WORDS + i];
    __syncthreads();
    azb_count(s_bits, s_cnt);
    for (int s = 0; s < steps; ++s) {
        if (s_meta[1]) break;
        unsigned r[4];
        az_philox4(d.seed, (unsigned)(d.slot_base + g), (unsigned)s_meta[2], r);
        const int k = (int)(((unsigned long long)r[0] * (unsigned)s_meta[3]) >> 32);
        const int pick = azb_kth_action(s_bits, s_cnt, k);
        if (lane == 0) s_pick = pick;
        __syncthreads();
        azb_play(d, g, s_pick, s_own, s_rows, s_tab, s_bits, s_cnt, s_used, s_last, s_meta, nullptr, nullptr);
    }
}

// Env::get_board: u8 planes [G][5][400]; possible_actions as a mask u8 [G][91*400]
extern "C" __global__ void azb_export_kernel(AzbDev d, uint8_t* planes, uint8_t* mask) {
    const int g = blockIdx.x;
    if (planes) {
        const int tm = d.to_move[g];
        for (int i = threadIdx.x; i < AZB_CELLS; i += blockDim.x) {
            const int y = i / AZB_N, x = i % AZB_N;
            for (int p = 0; p < 4; ++p) planes[((size_t)g * AZB_PLANES + p) * AZB_CELLS + i] = d.own[((size_t)g * 4 + p) * AZB_N + y] >> x & 1u;
            planes[((size_t)g * AZB_PLANES + 4) * AZB_CELLS + i] = (uint8_t)tm;
        }
    }
    if (mask)
        for (int a = threadIdx.x; a < AZB_ACTIONS; a += blockDim.x) {
            const int o = a / AZB_CELLS, cell = a % AZB_CELLS;
            mask[(size_t)g * AZB_ACTIONS + a] = d.legal_bits[(size_t)g * AZB_SET_WORDS + o * AZB_WORDS + cell / AZB_N] >> (cell % AZB_N) & 1u;
        }
}

// ------------------------------------------------------------------------------------------- search (MCTS)
// MCTS::search_probs for G Blokus games in lockstep rounds, one tree per wavefront, the reference's arithmetic
// (node.hpp:47-55 PUCT with int c_puct and a double intermediate; mcts.hpp:225-227 virtual loss; :328-345
// backup; :425-455 expansion with the raw prior and the 0.75/0.25 Dirichlet mix at a parentless root; :266-269
// terminal leaves) on what Blokus changes: four players, so an edge is valued for the player who chooses it
// (value = result[player to move at the parent], the reference's result[parent->player] with four entries),
// and an irregular action set, so a node carries its own child count: a node is a header slot {n = number of
// children, child = player to move} followed by its child edges in action order; an edge's link is the slot
// of its child's header (> 0), -1 a stub, -16 - code a terminal leaf whose reward is packed two bits per
// player in code.  A search starts from a fresh root, or -- az_blokus_search_reuse -- from what the kept tree holds
// for the position (azb_make_root_kernel).
#define AZB_TERMINAL(code_) (-16 - (code_))
__device__ __forceinline__ int azb_uni(int x) { return __builtin_amdgcn_readfirstlane(x); }

__device__ int azb_pack_reward(const float r[4]) {
    int code = 0;
    for (int q = 0; q < 4; ++q) code |= (r[q] > 0.f ? 1 : (r[q] < 0.f ? 2 : 0)) << (2 * q);
    return code;
}
__device__ void azb_unpack_reward(int code, float r[4]) {
    for (int q = 0; q < 4; ++q) { const int c = code >> (2 * q) & 3; r[q] = c == 1 ? 1.f : (c == 2 ? -1.f : 0.f); }
}

__device__ void azb_planes_to(uint8_t* pl, const uint32_t* s_own, int to_move) {
    for (int i = threadIdx.x; i < AZB_CELLS; i += 64) {
        const int y = i / AZB_N, x = i - AZB_N * y;
#pragma unroll
        for (int p = 0; p < 4; ++p) pl[p * AZB_CELLS + i] = s_own[p * AZB_N + y] >> x & 1u;
        pl[4 * AZB_CELLS + i] = (uint8_t)to_move;
    }
}
__device__ void azb_write_planes(const AzbDev& d, int slot, const uint32_t* s_own, int to_move) {
    azb_planes_to(d.leaf_planes + (size_t)slot * AZB_PLANES * AZB_CELLS, s_own, to_move);
}

// dirichlet(n_actions, alpha = 0.05) of util.h:42-58 drawn on the device for the root being expanded
__device__ void azb_device_noise(const AzbDev& d, int g, int count) {
    float* nz = d.noise + (size_t)g * AZB_NOISE_CAP;
    const unsigned ctr = d.noise_ctr[g];
    const unsigned stream = 0x80000000u | (unsigned)(d.slot_base + g);
    const int n = count < AZB_NOISE_CAP ? count : AZB_NOISE_CAP;
    double part = 0.0;
    for (int k = threadIdx.x; k < n; k += 64) {
        const float x = az_gamma_draw(d.seed, stream, (ctr * (unsigned)AZB_NOISE_CAP + (unsigned)k) * 32u, 0.05f);
        nz[k] = x;
        part += (double)x;
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
    __syncthreads();
    for (int k = threadIdx.x; k < n; k += 64) nz[k] = part > 0.0 ? (float)((double)nz[k] / part) : 1.0f / (float)n;
    if (threadIdx.x == 0) d.noise_ctr[g] = ctr + 1u;
    __syncthreads();
}

__device__ __forceinline__ AzEdge* azb_E(const AzbDev& d, int g) {
    return (__builtin_amdgcn_readfirstlane(d.arena[g]) ? d.edges1 : d.edges) + (size_t)g * d.edge_cap;
}
__device__ __forceinline__ uint16_t* azb_EA(const AzbDev& d, int g) {
    return (__builtin_amdgcn_readfirstlane(d.arena[g]) ? d.eact1 : d.eact) + (size_t)g * d.edge_cap;
}

// root = target (mcts.hpp:417-420): the subtree under the node whose header sits at slot `src` is copied into the
// other arena in BFS order, its root to slot 0, and becomes the game's tree (the reference drops its shared_ptr to
// the old root; here the old arena is simply left behind).  A node's place in the new arena is known the moment
// it is queued -- its header says how many children it has -- so links are rewritten on the fly.  Returns 0 if the
// subtree has more nodes than the queue holds (the caller then starts from a fresh root).
__device__ int azb_reroot(const AzbDev& d, int g, int src, int root_n) {
    const int lane = threadIdx.x;
    const int ar = __builtin_amdgcn_readfirstlane(d.arena[g]);
    const AzEdge* E0 = (ar ? d.edges1 : d.edges) + (size_t)g * d.edge_cap;
    const uint16_t* A0 = (ar ? d.eact1 : d.eact) + (size_t)g * d.edge_cap;
    AzEdge* E1 = (ar ? d.edges : d.edges1) + (size_t)g * d.edge_cap;
    uint16_t* A1 = (ar ? d.eact : d.eact1) + (size_t)g * d.edge_cap;
    int* queue = d.queue + (size_t)g * d.q_cap;
    if (lane == 0) queue[0] = src;
    __syncthreads();
    int new_count = 1, pos = 0;
    int alloc = 1 + __builtin_amdgcn_readfirstlane(E0[src].n);          // first free slot behind the new root's block
    bool fits = true;
    for (int i = 0; i < new_count && fits; ++i) {
        const int sh = __builtin_amdgcn_readfirstlane(queue[i]);
        const AzEdge h = E0[sh];
        const int cnt = __builtin_amdgcn_readfirstlane(h.n);
        if (lane == 0) { E1[pos] = h; A1[pos] = 0; }
        for (int c = 0; c < cnt; c += 64) {
            const int k = c + lane;
            const bool valid = k < cnt;
            AzEdge e = {0.f, 0, 0.f, -1};
            uint16_t act = 0;
            if (valid) { e = E0[sh + 1 + k]; act = A0[sh + 1 + k]; }
            const bool has = valid && e.child > 0;
            int sz = has ? 1 + E0[e.child].n : 0;                       // slots of the child's block
            const int own = sz;
            for (int o = 1; o < 64; o <<= 1) { const int y = __shfl_up(sz, o, 64); if (lane >= o) sz += y; }
            const unsigned long long m = __ballot(has);
            if (has) {
                const int rank = __popcll(m & ((1ull << lane) - 1ull));
                if (new_count + rank < d.q_cap) queue[new_count + rank] = e.child;
                e.child = alloc + sz - own;
            }
            if (valid) { E1[pos + 1 + k] = e; A1[pos + 1 + k] = act; }
            new_count += __popcll(m);
            alloc += __shfl(sz, 63, 64);
            if (new_count > d.q_cap) fits = false;
        }
        pos += 1 + cnt;
        __syncthreads();
    }
    if (!fits) return 0;
    if (lane == 0) { d.arena[g] = ar ^ 1; d.n_slots[g] = pos; d.n_nodes[g] = new_count; d.root_n[g] = root_n; }
    __syncthreads();
    return 1;
}

// __make_root (mcts.hpp:385-421) for the game's current position, then the search's counters.  find()
// (node.hpp:83-102) walks the kept tree by the moves played since its root: not found, or found as a child that
// was never expanded (a Node without state: searching from it is undefined in the reference) -> a fresh root;
// found below the root -> that subtree becomes the tree, with its statistics and without Dirichlet noise (its
// parent is not null, mcts.hpp:432); the root itself -> nothing changes, the visits keep accumulating.
extern "C" __global__ void __launch_bounds__(64) azb_make_root_kernel(AzbDev d, int budget) {
    const int g = blockIdx.x, lane = threadIdx.x;
    if (azb_uni(d.over[g])) { if (lane == 0) d.t_status[g] = AZ_ST_IDLE; return; }
    const int mv = azb_uni(d.moves[g]), rm = azb_uni(d.root_moves[g]);
    const uint16_t* H = d.hist + (size_t)g * AZB_MAX_MOVES;
    uint16_t* RH = d.root_hist + (size_t)g * AZB_MAX_MOVES;
    bool keep = false;
    if (azb_uni(d.n_nodes[g]) > 0 && rm >= 0 && rm <= mv) {
        bool mism = false;
        for (int t = lane; t < rm; t += 64) mism = mism || H[t] != RH[t];                   // node.hpp:89-90
        if (!__ballot(mism)) {
            if (rm == mv) keep = true;
            else {
                const AzEdge* E = azb_E(d, g);
                const uint16_t* EA = azb_EA(d, g);
                int hdr = 0, tn = 0;
                bool found = true;
                for (int t = rm; t < mv && found; ++t) {                                    // node.hpp:93-100
                    const int action = H[t];
                    const int cnt = azb_uni(E[hdr].n);
                    int at = -1;
                    for (int k = lane; k < cnt; k += 64) if (EA[hdr + 1 + k] == action) at = k;
                    for (int o = 32; o > 0; o >>= 1) at = max(at, __shfl_xor(at, o, 64));
                    at = azb_uni(at);
                    if (at < 0) { found = false; break; }
                    const AzEdge e = E[hdr + 1 + at];
                    const int child = azb_uni(e.child);
                    if (child <= 0) { found = false; break; }                               // stub or terminal
                    hdr = child; tn = azb_uni(e.n);
                }
                if (found) keep = azb_reroot(d, g, hdr, tn) != 0;
            }
        }
    }
    for (int t = lane; t < mv; t += 64) RH[t] = H[t];
    if (lane == 0) {
        if (!keep) { d.n_slots[g] = 0; d.n_nodes[g] = 0; d.root_n[g] = 0; }
        d.root_moves[g] = mv;
        d.sims_done[g] = 0; d.sims_target[g] = budget > 1 ? budget : 1;
        for (int k = 0; k < d.L; ++k) { d.pending[g * d.L + k] = AZ_PEND_NONE; d.path_len[g * d.L + k] = 0; }
        d.search_restarts[g] = 0;
        d.t_status[g] = AZ_ST_RUN;
    }
}

// Consume evaluation slot k of the previous round: expand (+ noise at the root) and back up.
__device__ void azb_tree_apply(const AzbDev& d, int g, int k) {
    const int lane = threadIdx.x;
    const int slot = g * d.L + k;
    const int pend = azb_uni(d.pending[slot]);
    if (pend == AZ_PEND_NONE) return;
    AzEdge* E = azb_E(d, g);
    uint16_t* EA = azb_EA(d, g);
    const int plen = azb_uni(d.path_len[slot]);
    const int* path = d.path + (size_t)slot * AZB_PATH_CAP;
    if (pend == AZ_PEND_ROOT || pend == AZ_PEND_LEAF) {
        const int cnt = azb_uni(d.leaf_count[slot]);
        const int hdr = azb_uni(d.n_slots[g]);
        if (hdr + 1 + cnt > d.edge_cap) {
            // Arena full.  The first time in a search: drop the tree and search the position again from a fresh root
            // (mcts.hpp:392-414).  If it fills up again the position is too wide for the arena at this budget: take
            // the virtual loss of the unfinished descent back and end the search with the simulations it has.
            // (The evaluations of this round that are still pending belong to the dropped tree and go with it; after a
            // truncation they are applied as usual, each giving up the same way if it does not fit either.)
            if (azb_uni(d.search_restarts[g]) == 0 && pend != AZ_PEND_ROOT) {
                if (lane == 0) {
                    d.n_slots[g] = 0; d.n_nodes[g] = 0; d.root_n[g] = 0; d.sims_done[g] = 0;
                    for (int kk = 0; kk < d.L; ++kk) d.pending[g * d.L + kk] = AZ_PEND_NONE;
                    d.restarts[g] += 1; d.search_restarts[g] = 1;
                }
            } else {
                const float vl = d.vl;
                for (int t = lane; t < plen; t += 64) {
                    AzEdge* e = &E[path[t]];
                    e->n = (int)((float)e->n - vl);
                    e->v = e->v + vl;
                }
                if (lane == 0) {
                    if (plen > 0) E[path[plen - 1]].child = -1;             // no longer waiting for an evaluation: a stub again
                    d.sims_target[g] = d.sims_done[g]; d.pending[slot] = AZ_PEND_NONE; d.truncated[g] += 1;
                }
            }
            __syncthreads();
            return;
        }
        const bool is_root = pend == AZ_PEND_ROOT;
        if (is_root && d.use_noise == 2) azb_device_noise(d, g, cnt);
        const float* pol = d.policy + (size_t)slot * AZB_ACTIONS;
        const float* nz = d.noise + (size_t)g * AZB_NOISE_CAP;
        const uint32_t* bits = d.leaf_bits + (size_t)slot * AZB_SET_WORDS;
        if (lane == 0) { AzEdge h = {0.f, cnt, 0.f, (int)d.leaf_player[slot]}; E[hdr] = h; EA[hdr] = 0; }
        int base = hdr + 1;
        for (int c = 0; c < AZB_SET_WORDS; c += 64) {                  // words in (o, y) order = action order
            const int i = c + lane;
            uint32_t w = i < AZB_SET_WORDS ? bits[i] : 0u;
            int x = __popc(w);
            const int own = x;
            for (int dd = 1; dd < 64; dd <<= 1) { const int y = __shfl_up(x, dd, 64); if (lane >= dd) x += y; }
            int k = base + x - own;
            while (w) {
                const int bit = __ffs((int)w) - 1;
                w &= w - 1u;
                const int action = i * AZB_N + bit;                    // (o*20 + y)*20 + x
                float prior = pol[action];
                if (is_root && d.use_noise) {                          // mcts.hpp:445 (double literals)
                    const int kk = k - (hdr + 1);
                    prior = (float)(0.75 * (double)prior + 0.25 * (double)(kk < AZB_NOISE_CAP ? nz[kk] : 0.f));
                }
                AzEdge e = {prior, 0, 0.f, -1};
                E[k] = e;
                EA[k] = (uint16_t)action;
                ++k;
            }
            base += __shfl(x, 63, 64);
        }
        if (lane == 0) {
            d.n_slots[g] = hdr + 1 + cnt;
            d.n_nodes[g] = d.n_nodes[g] + 1;
            if (!is_root) E[path[plen - 1]].child = hdr;               // leaf->add (mcts.hpp:316)
        }
        __syncthreads();
    }
    if (pend == AZ_PEND_ROOT) {
        if (lane == 0) d.root_n[g] = 1;                                 // mcts.hpp:412-413
    } else {                                                            // __backup (mcts.hpp:328-345)
        const float* res = pend == AZ_PEND_TERMINAL ? d.pend_result + 4 * slot : d.value + 4 * slot;
        const float vl = d.vl;
        for (int t = lane; t < plen; t += 64) {
            AzEdge* e = &E[path[t]];
            const float value = res[d.path_owner[(size_t)slot * AZB_PATH_CAP + t]];  // result[parent->player]
            e->n = (int)((float)e->n + (1.f - vl));
            e->v = e->v + (value + vl);
        }
        if (lane == 0) { d.root_n[g] = d.root_n[g] + 1; d.sims_done[g] = d.sims_done[g] + 1; }
    }
    if (lane == 0) d.pending[slot] = AZ_PEND_NONE;
    __syncthreads();
}

// A stub whose evaluation an earlier descent of this round is waiting for (between "not expanded", -1, and the terminal
// codes, <= -16).  The reference's searcher thread would block on that node's lock (mcts.hpp:224) until the other
// thread has expanded it; a descent that meets one here is taken back exactly and the round issues no more descents.
#define AZB_CHILD_WAITING (-2)

// One descent (__select_and_eval + _eval's env step, mcts.hpp:211-321) into evaluation slot k.  s_* hold the root
// position on entry (and the leaf's on return).  Returns 1 when the round's descents must stop: a fresh root, or a
// descent that met a waiting stub and was undone.
__device__ int azb_tree_select(const AzbDev& d, int g, int k, uint32_t* s_own, uint2* s_rows, const uint2* s_tab,
                               uint32_t* s_bits, int* s_cnt, uint32_t* s_used, int* s_last, int* s_meta,
                               int* s_path, int* s_oldn, float* s_oldv) {
    const int lane = threadIdx.x;
    const int slot = g * d.L + k;
    AzEdge* E = azb_E(d, g);
    const uint16_t* EA = azb_EA(d, g);
    int* path = d.path + (size_t)slot * AZB_PATH_CAP;
    uint8_t* owners = d.path_owner + (size_t)slot * AZB_PATH_CAP;
    if (azb_uni(d.n_nodes[g]) == 0) {                                   // fresh root: evaluate the root position
        for (int i = lane; i < AZB_SET_WORDS; i += 64) d.leaf_bits[(size_t)slot * AZB_SET_WORDS + i] = d.legal_bits[(size_t)g * AZB_SET_WORDS + i];
        azb_write_planes(d, slot, s_own, s_meta[0]);
        if (lane == 0) { d.leaf_count[slot] = s_meta[3]; d.leaf_player[slot] = (uint8_t)s_meta[0]; d.pending[slot] = AZ_PEND_ROOT; d.path_len[slot] = 0; }
        return 1;
    }
    const float vl = d.vl;
    const float cpf = (float)d.c_puct_i;
    int hdr = 0, par_n = azb_uni(d.root_n[g]), depth = 0, pend = AZ_PEND_NONE;
    float res[4] = {0.f, 0.f, 0.f, 0.f};
    for (;;) {
        const AzEdge h = E[hdr];
        const int cnt = azb_uni(h.n), owner = azb_uni(h.child);
        const double sq = sqrt((double)(par_n - 1));                    // node.hpp:54
        float best_val = -100.f;                                        // mcts.hpp:355
        int best_k = -1;
        for (int k = lane; k < cnt; k += 64) {
            const AzEdge e = E[hdr + 1 + k];
            const float cp = cpf * e.p;
            const float q = e.n != 0 ? __fdiv_rn(e.v, (float)e.n) : 0.f;
            const float val = (float)((double)q + (double)cp * sq / (double)(1 + e.n));
            if (val > best_val) { best_val = val; best_k = k; }
        }
        for (int o2 = 32; o2 > 0; o2 >>= 1) {                           // first maximum in child order
            const float ov = __shfl_xor(best_val, o2, 64);
            const int ok = __shfl_xor(best_k, o2, 64);
            if (ok >= 0 && (best_k < 0 || ov > best_val || (ov == best_val && ok < best_k))) { best_val = ov; best_k = ok; }
        }
        const int ei = hdr + 1 + azb_uni(best_k);
        AzEdge e = E[ei];
        const int action = EA[ei];
        if (lane == 0) { s_path[depth] = ei; s_oldn[depth] = e.n; s_oldv[depth] = e.v; }
        e.n = (int)((float)e.n + vl);                                   // mcts.hpp:225-227
        e.v = e.v - vl;
        if (lane == 0) { E[ei].n = e.n; E[ei].v = e.v; path[depth] = ei; owners[depth] = (uint8_t)owner; }
        azb_place(s_own, s_used, s_last, owner, action);
        ++depth;
        const int child = azb_uni(e.child);
        if (child > 0) { hdr = child; par_n = azb_uni(e.n); continue; }
        if (child == AZB_CHILD_WAITING) {                               // take the descent back: every edge of the path as it was
            __syncthreads();
            for (int t = lane; t < depth; t += 64) { E[s_path[t]].n = s_oldn[t]; E[s_path[t]].v = s_oldv[t]; }
            __syncthreads();
            return 1;
        }
        if (child <= -16) {                                             // stored terminal reward (mcts.hpp:232-233)
            azb_unpack_reward(-16 - child, res);
            pend = AZ_PEND_TERMINAL;
            break;
        }
        // stub: Env::step on the parent's state (mcts.hpp:261)
        int cnt2 = 0;
        const int next = azb_next_player(s_own, s_used, owner, s_rows, s_tab, s_bits, s_cnt, &cnt2);
        if (next < 0) {                                                 // terminal_add (mcts.hpp:266-269)
            azb_final_reward_dev(s_used, s_last, res);
            if (lane == 0) E[ei].child = AZB_TERMINAL(azb_pack_reward(res));
            pend = AZ_PEND_TERMINAL;
        } else {
            for (int i = lane; i < AZB_SET_WORDS; i += 64) d.leaf_bits[(size_t)slot * AZB_SET_WORDS + i] = s_bits[i];
            azb_write_planes(d, slot, s_own, next);
            if (lane == 0) { d.leaf_count[slot] = cnt2; d.leaf_player[slot] = (uint8_t)next; E[ei].child = AZB_CHILD_WAITING; }
            pend = AZ_PEND_LEAF;
        }
        break;
    }
    if (lane == 0) {
        d.pending[slot] = pend;
        d.path_len[slot] = depth;
        if (pend == AZ_PEND_TERMINAL) for (int q = 0; q < 4; ++q) d.pend_result[4 * slot + q] = res[q];
    }
    __syncthreads();
    return 0;
}

// One lockstep round: apply the pending evaluations in slot order, then (unless the search is complete) up to L
// descents, never more than the simulations still owed (mcts.hpp:161-174), each from the root position.
extern "C" __global__ void __launch_bounds__(64) azb_round_kernel(AzbDev d) {
    AZB_SMEM_DECL;
    __shared__ int s_path[AZB_PATH_CAP], s_oldn[AZB_PATH_CAP];
    __shared__ float s_oldv[AZB_PATH_CAP];
    const int g = blockIdx.x, lane = threadIdx.x;
    if (azb_uni(d.t_status[g]) != AZ_ST_RUN) return;
    for (int k = 0; k < d.L; ++k) azb_tree_apply(d, g, k);
    const int owed = azb_uni(d.sims_target[g]) - azb_uni(d.sims_done[g]);
    if (owed <= 0) {
        if (lane == 0) d.t_status[g] = AZ_ST_IDLE;
        return;
    }
    azb_load_table(s_tab);
    for (int k = 0; k < d.L && k < owed; ++k) {
        azb_load(d, g, s_own, s_used, s_last, s_meta);
        if (azb_tree_select(d, g, k, s_own, s_rows, s_tab, s_bits, s_cnt, s_used, s_last, s_meta, s_path, s_oldn, s_oldv)) break;
    }
}

extern "C" __global__ void azb_search_begin_kernel(AzbDev d, int budget) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    d.n_slots[g] = 0; d.n_nodes[g] = 0; d.root_n[g] = 0;
    d.sims_done[g] = 0; d.sims_target[g] = budget > 1 ? budget : 1;     // the loop body runs once even for budget <= 0
    for (int k = 0; k < d.L; ++k) { d.pending[g * d.L + k] = AZ_PEND_NONE; d.path_len[g * d.L + k] = 0; }
    d.search_restarts[g] = 0;
    d.t_status[g] = d.over[g] ? AZ_ST_IDLE : AZ_ST_RUN;
}

// Integer-hash pseudo network (parity evaluator): planes u8 [5*400] -> policy f32 [36400], value f32 [4].
extern "C" __global__ void __launch_bounds__(256) azb_eval_hash_kernel(const uint8_t* planes, float* policy, float* value) {
    __shared__ unsigned long long s_h[4];
    __shared__ unsigned s_sum[4];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint8_t* pl = planes + (size_t)g * AZB_PLANES * AZB_CELLS;
    unsigned long long H = 0;
    for (int i = tid; i < AZB_PLANES * AZB_CELLS; i += 256) H ^= az_splitmix64((unsigned long long)i * 256ull + pl[i]);
    for (int o = 32; o > 0; o >>= 1) H ^= __shfl_xor(H, o, 64);
    if (lane == 0) s_h[wv] = H;
    __syncthreads();
    H = s_h[0] ^ s_h[1] ^ s_h[2] ^ s_h[3];
    unsigned sum = 0;
    for (int i = tid; i < AZB_ACTIONS; i += 256) {
        const unsigned long long r = (az_splitmix64(H + (unsigned long long)i + 1ull) >> 40) & 0xFFFFull;
        sum += 1u + (unsigned)((r * r * r) >> 32);
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    if (lane == 0) s_sum[wv] = sum;
    __syncthreads();
    const float fs = (float)(s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3]);
    for (int i = tid; i < AZB_ACTIONS; i += 256) {
        const unsigned long long r = (az_splitmix64(H + (unsigned long long)i + 1ull) >> 40) & 0xFFFFull;
        policy[(size_t)g * AZB_ACTIONS + i] = __fdiv_rn((float)(1u + (unsigned)((r * r * r) >> 32)), fs);
    }
    if (tid < 4) {
        const int t = (int)((az_splitmix64(H ^ (0xABCDEFull + (unsigned long long)tid)) >> 40) & 0xFFFFull);
        value[4 * g + tid] = __fdiv_rn((float)t - 32768.0f, 40000.0f);
    }
}

// ----------------------------------------------------------------------------------------------- self-play
// The body of generate.cpp's move loop (:82-118) for one game whose search just finished: record the row
// (board planes + the search's policy), choose the move (sampled in proportion to the visits while the move
// number is below a fifth of the running mean game length, the first arg-max afterwards), play it, and at the
// end of the game hand the rows to the ring with the final reward and start the next game.  The policy of a
// row is kept sparse: (action, n_child / n_root) for the root's visited children in action order -- a dense
// [91,20,20] row is 146 KB of mostly zeros; a search of `budget` simulations visits at most `budget` children.
extern "C" __global__ void __launch_bounds__(64) azb_selfplay_move_kernel(AzbDev d) {
    AZB_SMEM_DECL;
    __shared__ int s_pick;
    const int g = blockIdx.x, lane = threadIdx.x;
    if (azb_uni(d.t_status[g]) == AZ_ST_RUN || azb_uni(d.over[g])) return;
    if (azb_uni(d.n_nodes[g]) == 0) {                                   // the root itself did not fit the arena: no move to play
        if (lane == 0) atomicAdd(&d.sp_counters[4], 1ull);
        return;
    }
    if (d.reuse && azb_uni(d.warm[g])) {                                // that was the warm-up search (generate.cpp:77): its
        if (lane == 0) d.warm[g] = 0;                                   // result is discarded, the tree stays
        return;
    }
    azb_load_table(s_tab);
    azb_load(d, g, s_own, s_used, s_last, s_meta);
    const AzEdge* E = azb_E(d, g);
    const uint16_t* EA = azb_EA(d, g);
    const int cnt = azb_uni(E[0].n);                                    // the root's children: slots 1 .. cnt
    const float rn = (float)azb_uni(d.root_n[g]);
    const int i = s_meta[2] + 1;                                        // generate.cpp:82, 1-based
    const int K = d.sp_k;
    const size_t row = (size_t)g * AZB_MAX_MOVES + (i - 1);
    azb_planes_to(d.row_planes + row * (AZB_PLANES * AZB_CELLS), s_own, s_meta[0]);
    {
        uint16_t* ra = d.row_act + row * K;
        float* rp = d.row_p + row * K;
        int base = 0;
        for (int c = 0; c < cnt; c += 64) {
            const int k = c + lane;
            const int n = k < cnt ? E[1 + k].n : 0;
            const unsigned long long m = __ballot(n > 0);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (n > 0 && pos < K) { ra[pos] = EA[1 + k]; rp[pos] = __fdiv_rn((float)n, rn); }     // mcts.hpp:136-146
            base += __popcll(m);
        }
        if (lane == 0) {
            d.row_cnt[row] = base < K ? base : K; d.row_tomove[row] = (uint8_t)s_meta[0];
            if (base > K) atomicAdd(&d.sp_counters[5], 1ull);
        }
    }
    int chosen = -1;
    if ((double)i < (double)d.sp_avg_len[g] * 0.2) {                   // generate.cpp:88: sample ~ visits
        unsigned part = 0;
        for (int k = lane; k < cnt; k += 64) part += (unsigned)E[1 + k].n;
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        unsigned r[4];
        const unsigned ctr = d.sp_sample_ctr[g];
        az_philox4(d.seed ^ 0x53414D50ull, (unsigned)(d.slot_base + g), ctr, r);
        const unsigned pick = (unsigned)(((unsigned long long)r[0] * part) >> 32);
        unsigned carry = 0;
        for (int c = 0; c < cnt && chosen < 0; c += 64) {
            const int k = c + lane;
            unsigned x = k < cnt ? (unsigned)E[1 + k].n : 0u;
            for (int o = 1; o < 64; o <<= 1) { const unsigned y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
            const unsigned long long m = __ballot(k < cnt && carry + x > pick);
            if (m) chosen = c + (__ffsll((long long)m) - 1);
            carry += __shfl(x, 63, 64);
        }
        if (lane == 0) d.sp_sample_ctr[g] = ctr + 1u;
    } else {                                                            // generate.cpp:95: first arg-max
        int bn = -1, bk = -1;
        for (int k = lane; k < cnt; k += 64) { const int n = E[1 + k].n; if (n > bn) { bn = n; bk = k; } }
        for (int o = 32; o > 0; o >>= 1) {
            const int on = __shfl_xor(bn, o, 64), ok = __shfl_xor(bk, o, 64);
            if (ok >= 0 && (bk < 0 || on > bn || (on == bn && ok < bk))) { bn = on; bk = ok; }
        }
        chosen = bk;
    }
    chosen = azb_uni(chosen);
    if (chosen < 0) chosen = 0;                                         // no visit anywhere: cannot happen once a search ran
    if (lane == 0) { s_pick = EA[1 + chosen]; atomicAdd(&d.sp_counters[2], 1ull); }
    __syncthreads();
    azb_play(d, g, s_pick, s_own, s_rows, s_tab, s_bits, s_cnt, s_used, s_last, s_meta, nullptr, nullptr);   // generate.cpp:103
    if (!s_meta[1]) return;
    // game over: buffer.send_reward(reward), stats, env.reset() (generate.cpp:110-118,75)
    __threadfence_block();                           // this move's row and the final reward were written by other lanes of the wave
    __syncthreads();
    int base = -1;
    if (lane == 0) {
        unsigned long long old = atomicAdd(&d.sp_counters[0], 0ull);
        for (;;) {
            if (old + (unsigned long long)i > (unsigned long long)d.ring_cap) { atomicAdd(&d.sp_counters[1], (unsigned long long)i); break; }
            const unsigned long long prev = atomicCAS(&d.sp_counters[0], old, old + (unsigned long long)i);
            if (prev == old) { base = (int)old; break; }
            old = prev;
        }
    }
    base = __shfl(base, 0, 64);
    if (base >= 0)
        for (int r = 0; r < i; ++r) {
            const size_t src = (size_t)g * AZB_MAX_MOVES + r, dst = (size_t)base + r;
            const uint32_t* sp = (const uint32_t*)(d.row_planes + src * (AZB_PLANES * AZB_CELLS));
            uint32_t* dp = (uint32_t*)(d.ring_planes + dst * (AZB_PLANES * AZB_CELLS));
            for (int c = lane; c < AZB_PLANES * AZB_CELLS / 4; c += 64) dp[c] = sp[c];
            const int n = d.row_cnt[src];
            for (int c = lane; c < n; c += 64) { d.ring_act[dst * K + c] = d.row_act[src * K + c]; d.ring_p[dst * K + c] = d.row_p[src * K + c]; }
            if (lane == 0) { d.ring_cnt[dst] = n; d.ring_tomove[dst] = d.row_tomove[src]; }
            if (lane < 4) d.ring_reward[dst * 4 + lane] = d.final_reward[g * 4 + lane];
        }
    if (lane == 0) {
        const long long gc = d.sp_game_count[g] + 1;                    // generate.cpp:25-29
        d.sp_game_count[g] = gc;
        const float al = d.sp_avg_len[g];
        d.sp_avg_len[g] = al + ((float)i - al) / (float)gc;
        atomicAdd(&d.sp_counters[3], 1ull);
        if (d.reuse) { d.root_moves[g] = -1; d.n_nodes[g] = 0; d.warm[g] = 1; }      // agent.clear() (generate.cpp:113); next: warm-up
    }
    azb_reset_game(d, g, s_own, s_used, s_rows, s_tab, s_bits, s_cnt, true);
}

extern "C" __global__ void azb_selfplay_init_kernel(AzbDev d) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    d.sp_avg_len[g] = 25.f;                                             // generate.cpp:69
    d.sp_game_count[g] = 0; d.sp_sample_ctr[g] = 0u;
    if (d.reuse) { d.warm[g] = 1; d.root_moves[g] = -1; d.n_nodes[g] = 0; }
}

// ------------------------------------------------------------------------------------------------ host
struct az_blokus {
    int device = 0;
    hipStream_t stream = nullptr;
    AzbDev d;
    std::vector<void*> allocs;
    int* d_actions = nullptr; float* d_reward = nullptr; uint8_t* d_done = nullptr; int* d_illegal = nullptr;
    uint8_t* d_export = nullptr;      // [G][91*400] mask or [G][5][400] planes
    int max_budget = 0;
    int64_t rounds = 0;
    hipEvent_t ev_tree = nullptr, ev_eval = nullptr;      // hand-offs between this stream and a tower context's
};

static int bfail(int code, const std::string& m) { return az_internal_fail(code, m.c_str()); }
#define BHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return bfail(AZ_E_HIP, std::string(#call " failed: ") + hipGetErrorString(e_)); } while (0)
#define BNEED(p) do { if (!(p)) return bfail(AZ_E_ARG, "null argument: " #p); } while (0)
#define AZB_CHK(call) do { int r_ = (call); if (r_ != AZ_OK) return r_; } while (0)

static const AzbOrient* host_table() {
    static AzbOrient T[AZB_ORIENTS + 8];
    static std::once_flag once;
    static int n = 0;
    std::call_once(once, [] { n = azb_build_orients(T); });
    return n == AZB_ORIENTS ? T : nullptr;
}

template <class T>
static int balloc(az_blokus* b, T** p, size_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, n * sizeof(T) < 16 ? 16 : n * sizeof(T)) != hipSuccess) return bfail(AZ_E_HIP, "hipMalloc failed");
    if (hipMemset(q, 0, n * sizeof(T) < 16 ? 16 : n * sizeof(T)) != hipSuccess) return bfail(AZ_E_HIP, "hipMemset failed");
    b->allocs.push_back(q);
    *p = (T*)q;
    return AZ_OK;
}

extern "C" int az_blokus_destroy(az_blokus* b) {
    if (!b) return AZ_OK;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (void* p : b->allocs) (void)hipFree(p);
    if (b->ev_tree) { (void)hipEventDestroy(b->ev_tree); (void)hipEventDestroy(b->ev_eval); }
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return AZ_OK;
}

extern "C" int az_blokus_create(az_blokus** out, int device, int G, uint64_t seed, int game_slot_base) {
    BNEED(out);
    *out = nullptr;
    if (G < 1) return bfail(AZ_E_ARG, "G must be >= 1");
    const AzbOrient* T = host_table();
    if (!T) return bfail(AZ_E_STATE, "orientation table does not have 91 entries");
    int ndev = 0;
    BHIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return bfail(AZ_E_HIP, "no HIP device: libazhip has no CPU fallback");
    if (device < 0 || device >= ndev) return bfail(AZ_E_ARG, "device out of range");
    BHIP(hipSetDevice(device));
    uint8_t size[AZB_PIECES];
    for (int o = 0; o < AZB_ORIENTS; ++o) size[T[o].piece] = T[o].n;
    BHIP(hipMemcpyToSymbol(HIP_SYMBOL(c_orient), T, sizeof(AzbOrient) * AZB_ORIENTS));
    BHIP(hipMemcpyToSymbol(HIP_SYMBOL(c_piece_size), size, sizeof size));
    az_blokus* b = new az_blokus;
    b->device = device;
    memset(&b->d, 0, sizeof b->d);
    b->d.G = G; b->d.seed = seed; b->d.slot_base = game_slot_base; b->d.L = 1;
    int rc = AZ_OK;
#define BA(ptr, n) do { if (rc == AZ_OK) rc = balloc(b, &(ptr), (size_t)(n)); } while (0)
    BA(b->d.own, (size_t)G * 4 * AZB_N); BA(b->d.used, (size_t)G * 4); BA(b->d.last_piece, (size_t)G * 4);
    BA(b->d.to_move, G); BA(b->d.over, G); BA(b->d.moves, G); BA(b->d.legal_count, G);
    BA(b->d.hist, (size_t)G * AZB_MAX_MOVES);
    BA(b->d.legal_bits, (size_t)G * AZB_SET_WORDS); BA(b->d.final_reward, (size_t)G * 4);
    BA(b->d_actions, G); BA(b->d_reward, (size_t)G * 4); BA(b->d_done, G); BA(b->d_illegal, 4);
    BA(b->d_export, (size_t)G * AZB_ACTIONS);
#undef BA
    if (rc == AZ_OK && hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) rc = bfail(AZ_E_HIP, "hipStreamCreate failed");
    if (rc == AZ_OK && hipDeviceSynchronize() != hipSuccess) rc = bfail(AZ_E_HIP, "hipDeviceSynchronize failed");   // zero fills (null stream) before the context's stream is used
    if (rc != AZ_OK) { const std::string keep = az_last_error(); az_blokus_destroy(b); return bfail(rc, keep); }
    *out = b;
    return az_blokus_reset(b);
}

extern "C" int az_blokus_reset(az_blokus* b) {
    BNEED(b);
    BHIP(hipSetDevice(b->device));
    hipLaunchKernelGGL(azb_reset_kernel, dim3(b->d.G), dim3(64), 0, b->stream, b->d);
    BHIP(hipGetLastError());
    BHIP(hipStreamSynchronize(b->stream));
    return AZ_OK;
}

extern "C" int az_blokus_step(az_blokus* b, const int32_t* actions, float* reward, uint8_t* done) {
    BNEED(b); BNEED(actions); BNEED(reward); BNEED(done);
    BHIP(hipSetDevice(b->device));
    const int G = b->d.G;
    BHIP(hipMemcpyAsync(b->d_actions, actions, sizeof(int) * G, hipMemcpyHostToDevice, b->stream));
    BHIP(hipMemsetAsync(b->d_illegal, 0, 16, b->stream));
    hipLaunchKernelGGL(azb_step_kernel, dim3(G), dim3(64), 0, b->stream, b->d, b->d_actions, b->d_reward, b->d_done, b->d_illegal);
    int illegal = 0;
    BHIP(hipMemcpyAsync(reward, b->d_reward, sizeof(float) * 4 * G, hipMemcpyDeviceToHost, b->stream));
    BHIP(hipMemcpyAsync(done, b->d_done, G, hipMemcpyDeviceToHost, b->stream));
    BHIP(hipMemcpyAsync(&illegal, b->d_illegal, sizeof(int), hipMemcpyDeviceToHost, b->stream));
    BHIP(hipStreamSynchronize(b->stream));
    if (illegal) return bfail(AZ_E_ARG, std::to_string(illegal) + " illegal action(s): not in the legal set of the player to move, or the game is over");
    return AZ_OK;
}

extern "C" int az_blokus_playout(az_blokus* b, int steps) {
    BNEED(b);
    if (steps < 0) return bfail(AZ_E_ARG, "steps is negative");
    BHIP(hipSetDevice(b->device));
    hipLaunchKernelGGL(azb_playout_kernel, dim3(b->d.G), dim3(64), 0, b->stream, b->d, steps);
    BHIP(hipGetLastError());
    return AZ_OK;
}

extern "C" int az_blokus_sync(az_blokus* b) { BNEED(b); BHIP(hipSetDevice(b->device)); BHIP(hipStreamSynchronize(b->stream)); return AZ_OK; }
extern "C" void* az_blokus_stream(az_blokus* b) { return b ? (void*)b->stream : nullptr; }

extern "C" int az_blokus_boards(az_blokus* b, uint8_t* planes) {
    BNEED(b); BNEED(planes);
    BHIP(hipSetDevice(b->device));
    hipLaunchKernelGGL(azb_export_kernel, dim3(b->d.G), dim3(256), 0, b->stream, b->d, b->d_export, (uint8_t*)nullptr);
    BHIP(hipMemcpyAsync(planes, b->d_export, (size_t)b->d.G * AZB_PLANES * AZB_CELLS, hipMemcpyDeviceToHost, b->stream));
    BHIP(hipStreamSynchronize(b->stream));
    return AZ_OK;
}

extern "C" int az_blokus_legal_mask(az_blokus* b, uint8_t* mask, int32_t* count) {
    BNEED(b);
    BHIP(hipSetDevice(b->device));
    if (mask) {
        hipLaunchKernelGGL(azb_export_kernel, dim3(b->d.G), dim3(256), 0, b->stream, b->d, (uint8_t*)nullptr, b->d_export);
        BHIP(hipMemcpyAsync(mask, b->d_export, (size_t)b->d.G * AZB_ACTIONS, hipMemcpyDeviceToHost, b->stream));
    }
    if (count) BHIP(hipMemcpyAsync(count, b->d.legal_count, sizeof(int) * b->d.G, hipMemcpyDeviceToHost, b->stream));
    BHIP(hipStreamSynchronize(b->stream));
    return AZ_OK;
}

extern "C" int az_blokus_state(az_blokus* b, uint8_t* to_move, uint8_t* over, int32_t* moves, uint32_t* used, float* final_reward) {
    BNEED(b);
    BHIP(hipSetDevice(b->device));
    const int G = b->d.G;
    if (to_move) BHIP(hipMemcpyAsync(to_move, b->d.to_move, G, hipMemcpyDeviceToHost, b->stream));
    if (over) BHIP(hipMemcpyAsync(over, b->d.over, G, hipMemcpyDeviceToHost, b->stream));
    if (moves) BHIP(hipMemcpyAsync(moves, b->d.moves, sizeof(int) * G, hipMemcpyDeviceToHost, b->stream));
    if (used) BHIP(hipMemcpyAsync(used, b->d.used, sizeof(uint32_t) * 4 * G, hipMemcpyDeviceToHost, b->stream));
    if (final_reward) BHIP(hipMemcpyAsync(final_reward, b->d.final_reward, sizeof(float) * 4 * G, hipMemcpyDeviceToHost, b->stream));
    BHIP(hipStreamSynchronize(b->stream));
    return AZ_OK;
}

// ------------------------------------------------------------------------------------------ search, host side
// MCTS(nthreads = 1, batch, vl, c_puct) for the Blokus games (mcts.hpp:41-73): allocates the trees.
extern "C" int az_blokus_search_config(az_blokus* b, float vl, float c_puct, int max_budget, int edge_cap) {
    BNEED(b);
    if (b->d.edges) return bfail(AZ_E_STATE, "the search is already configured");
    if (max_budget < 1) return bfail(AZ_E_ARG, "max_budget must be >= 1");
    // An arena that cannot hold the root's own block (1 header slot + one slot per legal action: 58 at the opening,
    // 660+ seen in mid-game positions) truncates every search before its first simulation, and a self-play game
    // on it would never advance: refuse such a size here (and az_blokus_selfplay_stats reports AZ_E_CAPACITY should a
    // root wider than the floor ever turn up).
    if (b->d.reuse && max_budget > 16000) return bfail(AZ_E_ARG, "max_budget too large for tree reuse");
    if (edge_cap > 0 && edge_cap < 1024 + max_budget)
        return bfail(AZ_E_ARG, "edge_cap " + std::to_string(edge_cap) + " cannot hold a root and its search: at least " +
                     std::to_string(1024 + max_budget) + " slots per game (1024 + max_budget), 0 = (max_budget + 2) * 1024");
    BHIP(hipSetDevice(b->device));
    const int G = b->d.G;
    b->d.vl = vl; b->d.c_puct_i = (int)c_puct;                        // Node::ucb takes int (node.hpp:47)
    // ~510 slots per node observed at 2048 games x 400 sims, 660+ at single roots; with reuse a tree carries the kept
    // subtree plus the new search (the first move's: the warm-up's and its own, generate.cpp:77)
    // (+ 64 nodes: at small budgets the kept subtree is large next to the budget)
    b->d.edge_cap = edge_cap > 0 ? edge_cap : (b->d.reuse ? 2 * max_budget + 64 : max_budget + 2) * 1024;
    b->max_budget = max_budget;
    int rc = AZ_OK;
#define BA(ptr, n) do { if (rc == AZ_OK) rc = balloc(b, &(ptr), (size_t)(n)); } while (0)
    BA(b->d.edges, (size_t)G * b->d.edge_cap); BA(b->d.eact, (size_t)G * b->d.edge_cap);
    BA(b->d.arena, G); BA(b->d.root_moves, G);
    if (b->d.reuse) {
        b->d.q_cap = 8 * max_budget + 64;                // nodes of a kept subtree; a larger one is dropped (fresh root)
        BA(b->d.edges1, (size_t)G * b->d.edge_cap); BA(b->d.eact1, (size_t)G * b->d.edge_cap);
        BA(b->d.root_hist, (size_t)G * AZB_MAX_MOVES); BA(b->d.queue, (size_t)G * b->d.q_cap); BA(b->d.warm, G);
    }
    BA(b->d.n_slots, G); BA(b->d.n_nodes, G); BA(b->d.root_n, G); BA(b->d.t_status, G); BA(b->d.sims_done, G); BA(b->d.sims_target, G);
    const size_t S = (size_t)G * b->d.L;                            // evaluation slots per round
    BA(b->d.pending, S); BA(b->d.path_len, S); BA(b->d.path, S * AZB_PATH_CAP); BA(b->d.path_owner, S * AZB_PATH_CAP);
    BA(b->d.pend_result, S * 4); BA(b->d.leaf_bits, S * AZB_SET_WORDS); BA(b->d.leaf_count, S); BA(b->d.leaf_player, S);
    BA(b->d.leaf_planes, S * AZB_PLANES * AZB_CELLS); BA(b->d.policy, S * AZB_ACTIONS); BA(b->d.value, S * 4);
    BA(b->d.noise, (size_t)G * AZB_NOISE_CAP); BA(b->d.restarts, G); BA(b->d.noise_ctr, G);
    BA(b->d.search_restarts, G); BA(b->d.truncated, G);
#undef BA
    if (rc != AZ_OK) { b->d.edges = nullptr; return rc; }
    BHIP(hipMemset(b->d.root_moves, 0xFF, sizeof(int) * G));         // -1: no tree
    BHIP(hipDeviceSynchronize());            // the zero fills ran on the null stream, the context's stream does not wait for it
    return AZ_OK;
}

// Before az_blokus_search_config: `leaves` descents per game and round, the counterpart of the reference's searcher
// threads on one tree (MCTS(nthreads, ...), mcts.hpp:54-72; the m,n,k search's az_config.leaves_per_round).  A round
// then evaluates G * leaves positions; a descent that reaches a leaf an earlier descent of the round is waiting on is
// taken back and ends the round's descents for that game (mcts.hpp:224: the thread would block there).
extern "C" int az_blokus_search_leaves(az_blokus* b, int leaves) {
    BNEED(b);
    if (b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_leaves must come before az_blokus_search_config (it sizes the evaluator's buffers)");
    if (leaves < 1 || leaves > 16) return bfail(AZ_E_ARG, "leaves must be 1..16");
    b->d.L = leaves;
    return AZ_OK;
}
extern "C" int az_blokus_eval_slots(const az_blokus* b) { return b ? b->d.G * b->d.L : 0; }

// Before az_blokus_search_config: keep each game's tree between searches, as the reference's MCTS object does
// (__make_root + find, mcts.hpp:385-421).  Costs a second arena per game.
extern "C" int az_blokus_search_reuse(az_blokus* b, int on) {
    BNEED(b);
    if (b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_reuse must come before az_blokus_search_config (it sizes the arenas)");
    b->d.reuse = on ? 1 : 0;
    return AZ_OK;
}

// MCTS::clear (mcts.hpp:462-465): drop every game's tree
extern "C" int az_blokus_tree_clear(az_blokus* b) {
    BNEED(b);
    if (!b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_config has not been called");
    BHIP(hipSetDevice(b->device));
    BHIP(hipMemsetAsync(b->d.root_moves, 0xFF, sizeof(int) * b->d.G, b->stream));
    BHIP(hipMemsetAsync(b->d.n_nodes, 0, sizeof(int) * b->d.G, b->stream));
    BHIP(hipMemsetAsync(b->d.n_slots, 0, sizeof(int) * b->d.G, b->stream));
    BHIP(hipMemsetAsync(b->d.root_n, 0, sizeof(int) * b->d.G, b->stream));
    return AZ_OK;
}

// util.h:42-58 made an input: noise f32 [G][n] (n <= 4096), entry k for the k-th legal action of a fresh root
extern "C" int az_blokus_set_noise(az_blokus* b, const float* noise, int n) {
    BNEED(b);
    if (!b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_config has not been called");
    BHIP(hipSetDevice(b->device));
    BHIP(hipStreamSynchronize(b->stream));
    if (!noise) { b->d.use_noise = 0; return AZ_OK; }
    if (n < 1 || n > AZB_NOISE_CAP) return bfail(AZ_E_ARG, "noise length out of range (1..4096)");
    BHIP(hipMemcpy2D(b->d.noise, sizeof(float) * AZB_NOISE_CAP, noise, sizeof(float) * n, sizeof(float) * n, b->d.G, hipMemcpyHostToDevice));
    b->d.use_noise = 1;
    return AZ_OK;
}

// The same noise drawn on the device (gamma(0.05) draws per legal action of every fresh root, normalised: util.h:42-58),
// a fresh draw per search; on = 0 switches noise off.
extern "C" int az_blokus_noise_device(az_blokus* b, int on) {
    BNEED(b);
    if (!b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_config has not been called");
    BHIP(hipSetDevice(b->device));
    BHIP(hipStreamSynchronize(b->stream));
    b->d.use_noise = on ? 2 : 0;
    return AZ_OK;
}

// The evaluator's device buffers, for an evaluator outside the library: planes u8 [G,5,20,20] in,
// policy f32 [G,91*400] and value f32 [G,4] out (the Evaluator's role, evaluator.hpp:80-124).
extern "C" int az_blokus_eval_io(az_blokus* b, uint8_t** planes_dev, float** policy_dev, float** value_dev) {
    BNEED(b);
    if (!b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_config has not been called");
    if (planes_dev) *planes_dev = b->d.leaf_planes;
    if (policy_dev) *policy_dev = b->d.policy;
    if (value_dev) *value_dev = b->d.value;
    return AZ_OK;
}

extern "C" int az_blokus_search_begin(az_blokus* b, int budget) {
    BNEED(b);
    if (!b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_config has not been called");
    if (budget > b->max_budget) return bfail(AZ_E_ARG, "budget exceeds the max_budget given to az_blokus_search_config");
    BHIP(hipSetDevice(b->device));
    if (b->d.reuse) hipLaunchKernelGGL(azb_make_root_kernel, dim3(b->d.G), dim3(64), 0, b->stream, b->d, budget);
    else hipLaunchKernelGGL(azb_search_begin_kernel, dim3((b->d.G + 63) / 64), dim3(64), 0, b->stream, b->d, budget);
    BHIP(hipGetLastError());
    return AZ_OK;
}

// One lockstep round of the search: apply the evaluations in the evaluator's buffers, select the next leaves
// (their planes go to the evaluator's input).  *n_running = games whose search is not finished yet (or NULL:
// no synchronisation).
extern "C" int az_blokus_search_round(az_blokus* b, int* n_running) {
    BNEED(b);
    if (!b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_config has not been called");
    BHIP(hipSetDevice(b->device));
    hipLaunchKernelGGL(azb_round_kernel, dim3(b->d.G), dim3(64), 0, b->stream, b->d);
    BHIP(hipGetLastError());
    b->rounds++;
    if (n_running) {
        std::vector<int> st((size_t)b->d.G);
        BHIP(hipMemcpyAsync(st.data(), b->d.t_status, sizeof(int) * b->d.G, hipMemcpyDeviceToHost, b->stream));
        BHIP(hipStreamSynchronize(b->stream));
        int run = 0;
        for (int x : st) run += x == AZ_ST_RUN;
        *n_running = run;
    }
    return AZ_OK;
}

extern "C" int az_blokus_eval_hash(az_blokus* b) {
    BNEED(b);
    if (!b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_config has not been called");
    BHIP(hipSetDevice(b->device));
    hipLaunchKernelGGL(azb_eval_hash_kernel, dim3(b->d.G * b->d.L), dim3(256), 0, b->stream, b->d.leaf_planes, b->d.policy, b->d.value);
    BHIP(hipGetLastError());
    return AZ_OK;
}

// MCTS::search_probs (mcts.hpp:105-155) for all G games with the integer-hash evaluator: `budget` simulations
// per game from a fresh root.
extern "C" int az_blokus_search(az_blokus* b, int budget) {
    AZB_CHK(az_blokus_search_begin(b, budget));
    const int bb = budget > 1 ? budget : 1;
    const int first = (bb + b->d.L - 1) / b->d.L + 2;                  // rounds before anybody is asked (L descents each at best)
    for (int r = 0; r < first; ++r) {
        AZB_CHK(az_blokus_search_round(b, nullptr));
        if (r < first - 1) AZB_CHK(az_blokus_eval_hash(b));
    }
    for (int extra = 0;; ++extra) {                                  // restarts (a full arena) need more rounds
        int run = 0;
        AZB_CHK(az_blokus_eval_hash(b));
        AZB_CHK(az_blokus_search_round(b, &run));
        if (!run) break;
        if (extra > 4 * (bb + 2)) return bfail(AZ_E_STATE, "search did not finish within its round bound");
    }
    return AZ_OK;
}

extern "C" int az_blokus_search_tower(az_blokus* b, az_ctx* tower, int budget) {
    BNEED(b); BNEED(tower);
    if (!b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_config has not been called");
    int N = 0, ci = 0, co = 0, vd = 0, mb = 0;
    AZB_CHK(az_tower_shape(tower, &N, &ci, &co, &vd, &mb));
    if (N != AZB_N || ci != AZB_PLANES || co != AZB_ORIENTS || vd != AZB_PLAYERS)
        return bfail(AZ_E_ARG, "the tower must be 20x20 with c_in 5, c_out 91, value_dim 4");
    const int S = b->d.G * b->d.L;
    if (mb < S) return bfail(AZ_E_ARG, "the tower context takes batches of " + std::to_string(mb) + ", a round of the search evaluates " + std::to_string(S) + " positions");
    // the tower's kernels read this handle's leaf planes and write its policy / value buffers in place
    if (az_ctx_device(tower) != b->device)
        return bfail(AZ_E_ARG, "the tower lives on GPU " + std::to_string(az_ctx_device(tower)) + ", the Blokus games on GPU " +
                     std::to_string(b->device) + ": they must share a GPU");
    BHIP(hipSetDevice(b->device));
    if (!b->ev_tree) { BHIP(hipEventCreateWithFlags(&b->ev_tree, hipEventDisableTiming)); BHIP(hipEventCreateWithFlags(&b->ev_eval, hipEventDisableTiming)); }
    hipStream_t ts = (hipStream_t)az_stream(tower);
    const bool dense = getenv("AZ_BLOKUS_DENSE_POLICY") != nullptr;     // A/B: the full [91,20,20] policy per leaf
    AZB_CHK(az_blokus_search_begin(b, budget));
    const int bb = budget > 1 ? budget : 1;
    const int first = (bb + b->d.L - 1) / b->d.L + 1;
    for (int r = 0, extra = 0;; ++r) {
        int run = 1;
        AZB_CHK(az_blokus_search_round(b, r < first ? nullptr : &run));           // past the expected end: ask who is still searching
        if (!run) break;
        if (r >= first && ++extra > 4 * (bb + 2)) return bfail(AZ_E_STATE, "search did not finish within its round bound");
        // the tree kernel's leaves -> the tower (its own stream) -> back: two event hand-offs, no host wait
        BHIP(hipEventRecord(b->ev_tree, b->stream));
        BHIP(hipStreamWaitEvent(ts, b->ev_tree, 0));
        // (only the priors the expansion will read: the leaf's legal actions, about 300 of the 36 400 entries)
        if (dense) AZB_CHK(az_tower_forward_dev(tower, b->d.leaf_planes, S, b->d.policy, b->d.value));
        else AZB_CHK(az_tower_forward_masked_dev(tower, b->d.leaf_planes, S, b->d.policy, b->d.value, b->d.leaf_bits));
        BHIP(hipEventRecord(b->ev_eval, ts));
        BHIP(hipStreamWaitEvent(b->stream, b->ev_eval, 0));
    }
    return AZ_OK;
}

// The root of game g after a search: its children in action order.  cap = 0 sizes (*n_children, *root_n).
extern "C" int az_blokus_search_result(az_blokus* b, int game, int cap, int32_t* n_children, int32_t* root_n,
                                       int32_t* actions, int32_t* n, float* v, float* p) {
    BNEED(b); BNEED(n_children);
    if (!b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_config has not been called");
    if (game < 0 || game >= b->d.G) return bfail(AZ_E_ARG, "game out of range");
    BHIP(hipSetDevice(b->device));
    BHIP(hipStreamSynchronize(b->stream));
    int nn = 0, rn = 0, st = 0, ar = 0;
    BHIP(hipMemcpy(&ar, b->d.arena + game, 4, hipMemcpyDeviceToHost));
    const AzEdge* const arena_e = (ar ? b->d.edges1 : b->d.edges) + (size_t)game * b->d.edge_cap;
    const uint16_t* const arena_a = (ar ? b->d.eact1 : b->d.eact) + (size_t)game * b->d.edge_cap;
    BHIP(hipMemcpy(&nn, b->d.n_nodes + game, 4, hipMemcpyDeviceToHost));
    BHIP(hipMemcpy(&rn, b->d.root_n + game, 4, hipMemcpyDeviceToHost));
    BHIP(hipMemcpy(&st, b->d.t_status + game, 4, hipMemcpyDeviceToHost));
    if (st == AZ_ST_RUN) return bfail(AZ_E_STATE, "the search of this game has not finished");
    AzEdge hdr = {0.f, 0, 0.f, 0};
    if (nn > 0) BHIP(hipMemcpy(&hdr, arena_e, sizeof hdr, hipMemcpyDeviceToHost));
    *n_children = nn > 0 ? hdr.n : 0;
    if (root_n) *root_n = rn;
    if (cap == 0) return AZ_OK;
    if (cap < *n_children) return bfail(AZ_E_ARG, "result buffers too small");
    BNEED(actions); BNEED(n); BNEED(v); BNEED(p);
    std::vector<AzEdge> ed((size_t)*n_children);
    std::vector<uint16_t> ea((size_t)*n_children);
    if (*n_children) {
        BHIP(hipMemcpy(ed.data(), arena_e + 1, sizeof(AzEdge) * ed.size(), hipMemcpyDeviceToHost));
        BHIP(hipMemcpy(ea.data(), arena_a + 1, 2 * ea.size(), hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < ed.size(); ++i) { actions[i] = ea[i]; n[i] = ed[i].n; v[i] = ed[i].v; p[i] = ed[i].p; }
    return AZ_OK;
}

// tallies of the search: expanded nodes and arena slots of the largest tree, restarts, rounds
extern "C" int az_blokus_search_stats(az_blokus* b, int32_t* max_nodes, int32_t* max_slots, int32_t* restarts, int64_t* rounds) {
    BNEED(b);
    if (!b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_config has not been called");
    BHIP(hipSetDevice(b->device));
    BHIP(hipStreamSynchronize(b->stream));
    const int G = b->d.G;
    std::vector<int> a((size_t)G), s((size_t)G), r((size_t)G), t((size_t)G);
    BHIP(hipMemcpy(a.data(), b->d.n_nodes, sizeof(int) * G, hipMemcpyDeviceToHost));
    BHIP(hipMemcpy(s.data(), b->d.n_slots, sizeof(int) * G, hipMemcpyDeviceToHost));
    BHIP(hipMemcpy(r.data(), b->d.restarts, sizeof(int) * G, hipMemcpyDeviceToHost));
    BHIP(hipMemcpy(t.data(), b->d.truncated, sizeof(int) * G, hipMemcpyDeviceToHost));
    int mn = 0, ms = 0, rs = 0;
    for (int g = 0; g < G; ++g) { mn = a[g] > mn ? a[g] : mn; ms = s[g] > ms ? s[g] : ms; rs += r[g] + t[g]; }
    if (max_nodes) *max_nodes = mn;
    if (max_slots) *max_slots = ms;
    if (restarts) *restarts = rs;
    if (rounds) *rounds = b->rounds;
    return AZ_OK;
}

// ---------------------------------------------------------------------------------------- self-play, host side
// generate.cpp:37-123 for the G games: az_blokus_selfplay_config allocates the per-game row buffers and a ring
// of `ring_rows` finished rows; az_blokus_selfplay_moves plays `n_moves` moves of every game (a search of
// `budget` simulations from a fresh root, then the move), starting the next game wherever one ends;
// az_blokus_replay_drain hands the finished rows over.
extern "C" int az_blokus_selfplay_config(az_blokus* b, int ring_rows) {
    BNEED(b);
    if (!b->d.edges) return bfail(AZ_E_STATE, "az_blokus_search_config has not been called");
    if (b->d.sp_counters) return bfail(AZ_E_STATE, "self-play is already configured");
    if (ring_rows < AZB_MAX_MOVES) return bfail(AZ_E_ARG, "ring_rows must hold at least one game (84 rows)");
    BHIP(hipSetDevice(b->device));
    // policy entries per row = visited root children: at most one per simulation ever made below the root -- the
    // budget from a fresh root; with reuse the kept subtree's too (the first move: 2 x budget, generate.cpp:77)
    const int G = b->d.G, K = b->d.reuse ? std::min(4 * b->max_budget, 4096) : b->max_budget;
    const size_t rows = (size_t)G * AZB_MAX_MOVES, R = (size_t)ring_rows;
    b->d.sp_k = K; b->d.ring_cap = ring_rows;
    int rc = AZ_OK;
#define BA(ptr, n) do { if (rc == AZ_OK) rc = balloc(b, &(ptr), (size_t)(n)); } while (0)
    BA(b->d.sp_avg_len, G); BA(b->d.sp_game_count, G); BA(b->d.sp_sample_ctr, G);
    BA(b->d.row_planes, rows * AZB_PLANES * AZB_CELLS); BA(b->d.row_tomove, rows); BA(b->d.row_cnt, rows);
    BA(b->d.row_act, rows * K); BA(b->d.row_p, rows * K);
    BA(b->d.ring_planes, R * AZB_PLANES * AZB_CELLS); BA(b->d.ring_tomove, R); BA(b->d.ring_cnt, R);
    BA(b->d.ring_act, R * K); BA(b->d.ring_p, R * K); BA(b->d.ring_reward, R * 4);
    BA(b->d.sp_counters, 8);
#undef BA
    if (rc != AZ_OK) { b->d.sp_counters = nullptr; return rc; }
    BHIP(hipDeviceSynchronize());            // the zero fills ran on the null stream
    hipLaunchKernelGGL(azb_selfplay_init_kernel, dim3((G + 63) / 64), dim3(64), 0, b->stream, b->d);
    BHIP(hipGetLastError());
    return AZ_OK;
}

// tower = NULL: the integer-hash evaluator
extern "C" int az_blokus_selfplay_moves(az_blokus* b, az_ctx* tower, int budget, int n_moves) {
    BNEED(b);
    if (!b->d.sp_counters) return bfail(AZ_E_STATE, "az_blokus_selfplay_config has not been called");
    if (n_moves < 0) return bfail(AZ_E_ARG, "n_moves is negative");
    for (int m = 0; m < n_moves; ++m) {
        AZB_CHK(tower ? az_blokus_search_tower(b, tower, budget) : az_blokus_search(b, budget));
        hipLaunchKernelGGL(azb_selfplay_move_kernel, dim3(b->d.G), dim3(64), 0, b->stream, b->d);
        BHIP(hipGetLastError());
    }
    return AZ_OK;
}

// Up to `cap` finished rows, oldest first, removed from the ring: planes u8 [cap][5*400], to_move u8 [cap],
// counts i32 [cap] policy entries per row, actions i32 / probs f32 [cap][K] (K = max_budget; entry j < counts[r]:
// policy[actions] = probs, every other entry of the [91,20,20] policy is zero), reward f32 [cap][4] by absolute
// player.  Returns the number of rows through *n_rows.
extern "C" int az_blokus_replay_drain(az_blokus* b, int cap, uint8_t* planes, uint8_t* to_move, int32_t* counts,
                                      int32_t* actions, float* probs, float* reward, int32_t* n_rows) {
    BNEED(b); BNEED(n_rows);
    *n_rows = 0;
    if (!b->d.sp_counters) return bfail(AZ_E_STATE, "az_blokus_selfplay_config has not been called");
    if (cap < 0) return bfail(AZ_E_ARG, "cap is negative");
    BHIP(hipSetDevice(b->device));
    BHIP(hipStreamSynchronize(b->stream));
    unsigned long long cnt = 0;
    BHIP(hipMemcpy(&cnt, b->d.sp_counters, 8, hipMemcpyDeviceToHost));
    const size_t n = (size_t)std::min<unsigned long long>(cnt, (unsigned long long)cap), K = (size_t)b->d.sp_k;
    const size_t PB = AZB_PLANES * AZB_CELLS;
    if (n == 0) return AZ_OK;
    BNEED(planes); BNEED(to_move); BNEED(counts); BNEED(actions); BNEED(probs); BNEED(reward);
    std::vector<uint16_t> act16(n * K);
    BHIP(hipMemcpy(planes, b->d.ring_planes, n * PB, hipMemcpyDeviceToHost));
    BHIP(hipMemcpy(to_move, b->d.ring_tomove, n, hipMemcpyDeviceToHost));
    BHIP(hipMemcpy(counts, b->d.ring_cnt, n * 4, hipMemcpyDeviceToHost));
    BHIP(hipMemcpy(act16.data(), b->d.ring_act, n * K * 2, hipMemcpyDeviceToHost));
    BHIP(hipMemcpy(probs, b->d.ring_p, n * K * 4, hipMemcpyDeviceToHost));
    BHIP(hipMemcpy(reward, b->d.ring_reward, n * 16, hipMemcpyDeviceToHost));
    for (size_t r = 0; r < n; ++r)
        for (size_t j = 0; j < K; ++j) {
            const bool live = (int)j < counts[r];
            actions[r * K + j] = live ? act16[r * K + j] : -1;
            if (!live) probs[r * K + j] = 0.f;
        }
    const size_t rest = (size_t)cnt - n;
    // keep the tail: shift it to the front, n rows at a time from the low end (source and destination of one copy never overlap)
    for (size_t off = 0; off < rest; off += n) {
        const size_t k = std::min(n, rest - off);
        BHIP(hipMemcpy(b->d.ring_planes + off * PB, b->d.ring_planes + (off + n) * PB, k * PB, hipMemcpyDeviceToDevice));
        BHIP(hipMemcpy(b->d.ring_tomove + off, b->d.ring_tomove + off + n, k, hipMemcpyDeviceToDevice));
        BHIP(hipMemcpy(b->d.ring_cnt + off, b->d.ring_cnt + off + n, k * 4, hipMemcpyDeviceToDevice));
        BHIP(hipMemcpy(b->d.ring_act + off * K, b->d.ring_act + (off + n) * K, k * K * 2, hipMemcpyDeviceToDevice));
        BHIP(hipMemcpy(b->d.ring_p + off * K, b->d.ring_p + (off + n) * K, k * K * 4, hipMemcpyDeviceToDevice));
        BHIP(hipMemcpy(b->d.ring_reward + off * 4, b->d.ring_reward + (off + n) * 4, k * 16, hipMemcpyDeviceToDevice));
    }
    const unsigned long long left = rest;
    BHIP(hipMemcpy(b->d.sp_counters, &left, 8, hipMemcpyHostToDevice));
    *n_rows = (int)n;
    return AZ_OK;
}

// moves played, games finished, rows waiting in the ring, rows dropped because the ring was full
extern "C" int az_blokus_selfplay_stats(az_blokus* b, int64_t* moves, int64_t* games, int64_t* rows, int64_t* dropped) {
    BNEED(b);
    if (!b->d.sp_counters) return bfail(AZ_E_STATE, "az_blokus_selfplay_config has not been called");
    BHIP(hipSetDevice(b->device));
    BHIP(hipStreamSynchronize(b->stream));
    unsigned long long k[6];
    BHIP(hipMemcpy(k, b->d.sp_counters, sizeof k, hipMemcpyDeviceToHost));
    if (rows) *rows = (int64_t)k[0];
    if (dropped) *dropped = (int64_t)k[1];
    if (moves) *moves = (int64_t)k[2];
    if (games) *games = (int64_t)k[3];
    if (k[4]) return bfail(AZ_E_CAPACITY, std::to_string(k[4]) + " moves were not played: a game's root (1 + its legal actions) does not fit the " +
                           std::to_string(b->d.edge_cap) + "-slot arena (az_blokus_search_config edge_cap)");
    if (k[5]) return bfail(AZ_E_CAPACITY, std::to_string(k[5]) + " rows had more visited root children than the " + std::to_string(b->d.sp_k) +
                           " policy entries a row holds: the rest of their policy was cut off");
    return AZ_OK;
}

// policy entries a self-play row holds (the K of az_blokus_replay_drain's actions / probs), 0 before az_blokus_selfplay_config
extern "C" int az_blokus_selfplay_row_cap(const az_blokus* b) { return b ? b->d.sp_k : 0; }

// ------------------------------------------------------------------------- host twin (no GPU needed)
extern "C" int az_blokus_orient(int o, int* piece, int* h, int* w, int8_t* cells_yx) {
    const AzbOrient* T = host_table();
    if (!T || o < 0 || o >= AZB_ORIENTS || !piece || !h || !w || !cells_yx) return bfail(AZ_E_ARG, "bad argument to az_blokus_orient");
    *piece = T[o].piece; *h = T[o].h; *w = T[o].w;
    for (int i = 0; i < T[o].n; ++i) { cells_yx[2 * i] = T[o].cy[i]; cells_yx[2 * i + 1] = T[o].cx[i]; }
    return T[o].n;
}
extern "C" size_t az_blokus_host_state_size(void) { return sizeof(AzbGame); }
extern "C" int az_blokus_host_reset(void* state) { BNEED(state); azb_reset((AzbGame*)state); return AZ_OK; }
extern "C" int az_blokus_host_step(void* state, int action, float* reward) {
    BNEED(state); BNEED(reward);
    const int r = azb_step((AzbGame*)state, host_table(), action, reward);
    return r < 0 ? bfail(AZ_E_ARG, "illegal action") : r;
}
extern "C" int az_blokus_host_legal(const void* state, uint8_t* mask) {
    BNEED(state);
    const AzbGame* g = (const AzbGame*)state;
    int n = 0;
    for (int a = 0; a < AZB_ACTIONS; ++a) {
        const bool ok = !g->over && azb_legal(g, host_table(), g->to_move, a);
        if (mask) mask[a] = ok;
        n += ok;
    }
    return n;
}
extern "C" int az_blokus_host_info(const void* state, uint8_t* cells, int* to_move, int* over, uint32_t* used, int32_t* scores) {
    BNEED(state);
    const AzbGame* g = (const AzbGame*)state;
    if (cells) memcpy(cells, g->cells, AZB_CELLS);
    if (to_move) *to_move = g->to_move;
    if (over) *over = g->over;
    if (used) for (int p = 0; p < 4; ++p) used[p] = g->used[p];
    if (scores) { int s[4]; azb_scores(g, host_table(), s); for (int p = 0; p < 4; ++p) scores[p] = s[p]; }
    return AZ_OK;
}
