// tree.hip -- lockstep environment + MCTS kernels, one game (one tree) per
// 64-lane wavefront, trees in flat HBM arenas.
//
// Restates, with bit-exact visit statistics (L = 1 leaf per tree per round is
// the reference at nthreads = 1; L > 1 is its nthreads = L in one defined order):
//   MCTS::__select_and_eval / __choose / Node::ucb  (mcts.hpp:211-237,350-381; node.hpp:47-55)
//   MCTS::_eval / _append_children                   (mcts.hpp:240-321,425-455)
//   MCTS::__backup                                   (mcts.hpp:323-346)
//   MCTS::__make_root / find                         (mcts.hpp:385-421; node.hpp:83-102)
//   MCTS::search_probs output policy                 (mcts.hpp:136-146)
//   generate.cpp's per-move / per-game loop          (generate.cpp:72-120)
//   Env::step / possible_actions / get_board         (call sites mcts.hpp:251-271,427)
// Built with -ffp-contract=off: every float/double expression keeps the
// reference's evaluation order and promotions.
#include "az_dev.h"

#define AZ_WAVE 64
// Optional phase timing (build with -DAZ_TREE_TIMING): block 7 stamps the 100 MHz clock.
#ifdef AZ_TREE_TIMING
__device__ unsigned long long az_tree_ts[32];
#define TTS(slot_) do { if (blockIdx.x == 7 && threadIdx.x == 0) az_tree_ts[slot_] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" int az_tree_timing_read(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(az_tree_ts), sizeof(unsigned long long) * 32);
}
#else
#define TTS(slot_) do {} while (0)
#endif
#define AZ_CELLS_MAX 512
// LDS of a tree block: board bytes, then per-descent records (old n, old v, action, edge), sized by the board: the
// records hold A + 1 plies at most.  7.5 KB at 21x21; 3.9 KB at 15x15, where a CU holds two such blocks beside a
// resident tower workgroup (8 KB are left): should two tree waves of the persistent pair ever land on one CU -- the
// driver takes the waves of a process off the GPU and puts them back, in any order, whenever the process makes or
// drops a hardware queue -- the tower workgroup of that CU still fits.
#define AZ_PATH_MAX 448          // plies of one descent: A + 1 <= 442
__host__ __device__ __forceinline__ int az_smem_cells(int A) { return (A + 15) & ~15; }
__host__ __device__ __forceinline__ int az_smem_plies(int A) { return (A + 1 + 7) & ~7; }
__host__ __device__ __forceinline__ int az_smem_bytes(int A) { return az_smem_cells(A) + 4 * az_smem_plies(A) * 4; }

__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
// phase clocks of the verbosity-3 report (mcts.hpp:187-201,248-303): read only while the context asks for them
#define AZ_TICK(d_) ((d_).timing ? __builtin_amdgcn_s_memrealtime() : 0ull)
#define AZ_TOCK(d_, g_, slot_, t0_) do { if ((d_).timing && threadIdx.x == 0) \
        (d_).g_ctr[(size_t)(g_) * AZ_GC_STRIDE + (slot_)] += __builtin_amdgcn_s_memrealtime() - (t0_); } while (0)
__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << threadIdx.x) - 1ull;
}

// ---------------------------------------------------------------- env rules
// K-in-a-row through the stone just placed at `action`: lanes 0..4*(2K-1)-1
// each test one cell of one of the four lines, one ballot gathers them.
__device__ bool az_win_check(const uint8_t* s_cells, int N, int K, int action, int stone) {
    const int span = 2 * K - 1;
    const int l = threadIdx.x;
    bool hit = false;
    if (l < 4 * span) {
        const int dir = l / span, off = l % span - (K - 1);
        const int dy = dir == 0 ? 0 : 1;
        const int dx = dir == 0 ? 1 : (dir == 1 ? 0 : (dir == 2 ? 1 : -1));
        const int y = action / N + off * dy, x = action % N + off * dx;
        hit = y >= 0 && y < N && x >= 0 && x < N && s_cells[y * N + x] == stone;
    }
    const unsigned long long m = __ballot(hit);
    bool win = false;
    for (int d = 0; d < 4; ++d) {
        const unsigned bits = (unsigned)(m >> (d * span)) & ((1u << span) - 1u);
        int run = 1;
        for (int s = 1; s < K; ++s) { if ((bits >> (K - 1 + s)) & 1u) ++run; else break; }
        for (int s = 1; s < K; ++s) { if ((bits >> (K - 1 - s)) & 1u) ++run; else break; }
        win = win || run >= K;
    }
    return win;
}

__device__ void az_write_planes(const AzTreeDev& d, int slot, const uint8_t* s_cells, int to_move) {
    uint8_t* pl = d.leaf_planes + (size_t)slot * 3 * d.A;
    for (int i = threadIdx.x; i < d.A; i += AZ_WAVE) {
        const uint8_t c = s_cells[i];
        pl[i] = c == 1;
        pl[d.A + i] = c == 2;
        pl[2 * d.A + i] = (uint8_t)to_move;
    }
}

// ------------------------------------------------------- Dirichlet on device
// util.h:42-58: gamma(alpha) draws (az_gamma_draw, az_dev.h), double sum, divide.
__device__ void az_device_noise(const AzTreeDev& d, int g, int count) {
    float* nz = d.noise + (size_t)g * d.A;
    const unsigned ctr = d.noise_ctr[g];
    const unsigned stream = 0x80000000u | (unsigned)(d.slot_base + g);
    double part = 0.0;
    for (int k = threadIdx.x; k < count; k += AZ_WAVE) {
        const float x = az_gamma_draw(d.seed, stream, (ctr * 512u + (unsigned)k) * 32u, 0.05f);
        nz[k] = x;
        part += (double)x;
    }
    for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, AZ_WAVE);
    __syncthreads();
    for (int k = threadIdx.x; k < count; k += AZ_WAVE)
        nz[k] = part > 0.0 ? (float)((double)nz[k] / part) : 1.0f / (float)count;
    if (threadIdx.x == 0) d.noise_ctr[g] = ctr + 1u;
    __syncthreads();
}

// high-water marks of the arena, recorded whenever a tree is about to shrink
__device__ __forceinline__ void az_note_high_water(const AzTreeDev& d, int g) {
    d.g_hw_nodes[g] = max(d.g_hw_nodes[g], d.n_nodes[g]);
    d.g_hw_edges[g] = max(d.g_hw_edges[g], d.n_edges[g]);
}

// ------------------------------------------------------------ re-root (find)
// Copies the subtree whose root block starts at edge `target_fe` into the other arena in
// BFS order, so that the new root's block starts at edge 0 (the reference just drops the
// shared_ptr to the old root, mcts.hpp:417-420).  A node is its block of child edges; an
// expanded node at BFS level l below the new root has A - hist_len - l of them (one per
// empty cell), so block sizes need no storage and a child's place in the new arena is
// known the moment it is queued.
__device__ void az_reroot(const AzTreeDev& d, int g, int target_fe, int target_n) {
    const int ar = uni(d.arena[g]);
    const AzEdge* E0 = d.edges[ar] + (size_t)g * d.edge_cap;
    const uint16_t* A0 = d.eact[ar] + (size_t)g * d.edge_cap;
    AzEdge* E1 = d.edges[ar ^ 1] + (size_t)g * d.edge_cap;
    uint16_t* A1 = d.eact[ar ^ 1] + (size_t)g * d.edge_cap;
    // BFS queue of old block starts in HBM (the block's LDS stays small enough to share a CU
    // with a resident tower workgroup); the wave reads it 64 entries at a time
    int* queue = d.queue + (size_t)g * d.node_cap;
    const int ne_root = d.A - uni(d.hist_len[g]);
    if (threadIdx.x == 0) queue[0] = target_fe;
    __syncthreads();
    int new_count = 1, new_edges = 0;
    int level = 0, level_end = 1, alloc = ne_root;                      // alloc: first free edge of the new arena
    int win_base = 0, win_end = 0, win_val = 0;                         // queue[win_base + lane] for entries < win_end
    for (int i = 0; i < new_count; ++i) {
        if (i == level_end) { ++level; level_end = new_count; }
        if (i >= win_end) {
            win_base = i;
            win_end = min(i + AZ_WAVE, new_count);
            win_val = win_base + (int)threadIdx.x < win_end ? queue[win_base + threadIdx.x] : 0;
        }
        const int fe = __shfl(win_val, i - win_base, AZ_WAVE);
        const int ne = ne_root - level, ne_child = ne - 1;
        for (int c = 0; c < ne; c += AZ_WAVE) {
            const int k = c + threadIdx.x;
            const bool valid = k < ne;
            AzEdge e = {0.f, 0, 0.f, -1};
            uint16_t act = 0;
            if (valid) { e = E0[fe + k]; act = A0[fe + k]; }
            const bool has = valid && e.child >= 0;
            const unsigned long long m = __ballot(has);
            if (has) {
                const int rank = __popcll(m & lanemask_lt());
                queue[new_count + rank] = e.child;
                e.child = alloc + rank * ne_child;
            }
            if (valid) { E1[new_edges + k] = e; A1[new_edges + k] = act; }
            new_count += __popcll(m);
            alloc += __popcll(m) * ne_child;
        }
        new_edges += ne;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        az_note_high_water(d, g);
        d.arena[g] = ar ^ 1;
        d.n_nodes[g] = new_count;
        d.n_edges[g] = new_edges;
        d.root_n[g] = target_n;
        const int hl = d.hist_len[g];
        for (int t = 0; t < hl; ++t) d.root_hist[(size_t)g * d.A + t] = d.hist[(size_t)g * d.A + t];
        d.root_hist_len[g] = hl;
    }
    __syncthreads();
}

__device__ void az_drop_tree(const AzTreeDev& d, int g) {
    if (threadIdx.x == 0) { az_note_high_water(d, g); d.root_hist_len[g] = -1; d.n_nodes[g] = 0; d.n_edges[g] = 0; d.root_n[g] = 0; }
    __syncthreads();
}

// find() + the reuse rule of __make_root for the current env history.
// returns 1 if the position is searchable, 0 if the root is a finished game.
__device__ int az_make_root(const AzTreeDev& d, int g) {
    const int hl = uni(d.hist_len[g]);
    const int rhl = uni(d.root_hist_len[g]);
    if (uni(d.game_over[g])) { az_drop_tree(d, g); return 0; }
    if (rhl < 0) return 1;
    if (hl < rhl) { az_drop_tree(d, g); return 1; }                    // node.hpp:86
    const uint16_t* H = d.hist + (size_t)g * d.A;
    const uint16_t* RH = d.root_hist + (size_t)g * d.A;
    bool mism = false;
    for (int t = threadIdx.x; t < rhl; t += AZ_WAVE) mism = mism || H[t] != RH[t];
    if (__ballot(mism)) { az_drop_tree(d, g); return 1; }              // node.hpp:89-90
    if (hl == rhl) return 1;                                           // target == root: keep accumulating
    const int ar = uni(d.arena[g]);
    const AzEdge* E = d.edges[ar] + (size_t)g * d.edge_cap;
    const uint16_t* EA = d.eact[ar] + (size_t)g * d.edge_cap;
    int fe = 0, tn = 0;                                                // the root's block starts at edge 0
    for (int t = rhl; t < hl; ++t) {                                   // node.hpp:93-100
        const int action = H[t];
        const int ne = d.A - t;                                        // one child per empty cell after t plies
        int found = -1;
        for (int k = threadIdx.x; k < ne; k += AZ_WAVE)
            if (EA[fe + k] == action) found = k;
        for (int o = 32; o > 0; o >>= 1) found = max(found, __shfl_xor(found, o, AZ_WAVE));
        found = uni(found);
        if (found < 0) { az_drop_tree(d, g); return 1; }
        const AzEdge e = E[fe + found];
        const int child = uni(e.child);
        // a never-visited stub has no state: searching from it is undefined in
        // the reference; a terminal child cannot be current while !game_over
        if (child < 0) { az_drop_tree(d, g); return 1; }
        fe = child;
        tn = uni(e.n);
    }
    az_reroot(d, g, fe, tn);
    return 1;
}

// -------------------------------------------------------------- apply phase
// Consumes evaluation slot k of the previous round: expand (+Dirichlet at a
// fresh root) and back up.
__device__ void az_apply(const AzTreeDev& d, int g, int k, uint8_t* s_cells, int* s_act) {
    const int slot = g * d.L + k;
    const int pend = uni(d.pending[slot]);
    if (pend == AZ_PEND_NONE) return;
    const int A = d.A;
    const int ar = uni(d.arena[g]);
    AzEdge* E = d.edges[ar] + (size_t)g * d.edge_cap;
    uint16_t* EA = d.eact[ar] + (size_t)g * d.edge_cap;
    const int plen = uni(d.path_len[slot]);
    const int* path = d.path + (size_t)slot * d.path_cap;
    const int rt = uni(d.to_move[g]);
    const int hl = uni(d.hist_len[g]);
    TTS(8);
    const unsigned long long tk_a = AZ_TICK(d);

    if (pend == AZ_PEND_ROOT || pend == AZ_PEND_LEAF) {
        // the leaf position = root position + the moves along the path (one lane per ply:
        // the two dependent loads of every ply are in flight together)
        for (int t = threadIdx.x; t < plen; t += AZ_WAVE) {
            const int act = EA[path[t]];
            s_act[t] = act;
            s_cells[act] = (uint8_t)((rt ^ (t & 1)) + 1);
        }
        __syncthreads();
        const int nn = uni(d.n_nodes[g]), ne0 = uni(d.n_edges[g]);
        const int legal = A - (hl + plen);
        if (nn >= d.node_cap || ne0 + legal > d.edge_cap) {
            // The arena is full (the reference's heap has no such limit).  Give the tree up and search this
            // position again from a fresh root, which is what __make_root does whenever find() misses
            // (mcts.hpp:392-414): the slot keeps playing, the event is tallied per game.  Evaluations of this
            // round that are still pending belong to the dropped tree and are discarded with it.
            for (int t = threadIdx.x; t < plen; t += AZ_WAVE) s_cells[s_act[t]] = 0;
            __syncthreads();
            az_drop_tree(d, g);
            if (threadIdx.x == 0) {
                d.sims_done[g] = 0;
                for (int kk = 0; kk < d.L; ++kk) d.pending[g * d.L + kk] = AZ_PEND_NONE;
                d.g_ctr[(size_t)g * AZ_GC_STRIDE + AZ_GC_RESTARTS] += 1ull;
            }
            __syncthreads();
            return;
        }
        const bool is_root = pend == AZ_PEND_ROOT;
        if (is_root && d.noise_mode == 1) az_device_noise(d, g, legal);
        const float* pol = d.policy + (size_t)slot * A;
        const float* nz = d.noise + (size_t)g * A;
        int base = 0;
        for (int c = 0; c < A; c += AZ_WAVE) {                          // mcts.hpp:438-454
            const int cell = c + threadIdx.x;
            const bool lg = cell < A && s_cells[cell] == 0;
            const unsigned long long m = __ballot(lg);
            if (lg) {
                const int k = base + __popcll(m & lanemask_lt());
                float prior = pol[cell];
                if (is_root)                                            // mcts.hpp:445 (double literals)
                    prior = (float)(0.75 * (double)prior + 0.25 * (double)nz[k]);
                AzEdge e = {prior, 0, 0.f, -1};
                E[ne0 + k] = e;
                EA[ne0 + k] = (uint16_t)cell;
            }
            base += __popcll(m);
        }
        if (threadIdx.x == 0) {
            d.n_nodes[g] = nn + 1;
            d.n_edges[g] = ne0 + base;
            d.g_ctr[(size_t)g * AZ_GC_STRIDE + AZ_GC_EXPANDED] += (unsigned long long)base;
            if (!is_root) E[path[plen - 1]].child = ne0;                // leaf->add (mcts.hpp:316): link = its block's first edge
        }
        __syncthreads();
        // restore the root position in LDS
        for (int t = threadIdx.x; t < plen; t += AZ_WAVE) s_cells[s_act[t]] = 0;
        __syncthreads();
    }
    TTS(9);
    if (pend != AZ_PEND_TERMINAL) AZ_TOCK(d, g, AZ_GC_T_APPEND, tk_a);   // _append_children (mcts.hpp:300-303)
    const unsigned long long tk_b = AZ_TICK(d);
    if (pend == AZ_PEND_ROOT) {                                         // mcts.hpp:412-413
        if (threadIdx.x == 0) {
            d.root_n[g] = 1;
            for (int t = 0; t < hl; ++t) d.root_hist[(size_t)g * A + t] = d.hist[(size_t)g * A + t];
            d.root_hist_len[g] = hl;
        }
    } else {                                                            // __backup, mcts.hpp:328-345
        const float* res = pend == AZ_PEND_TERMINAL ? d.pend_result + 2 * slot : d.value + 2 * slot;
        const float r0 = res[0], r1 = res[1];
        const float vl = d.vl;
        for (int t = threadIdx.x; t < plen; t += AZ_WAVE) {
            AzEdge* e = &E[path[t]];
            const int pp = rt ^ (t & 1);                                // parent->player
            const float value = pp ? r1 : r0;
            e->n = (int)((float)e->n + (1.f - vl));
            e->v = e->v + (value + vl);
        }
        if (threadIdx.x == 0) {
            d.root_n[g] = d.root_n[g] + 1;
            d.sims_done[g] = d.sims_done[g] + 1;
        }
    }
    if (threadIdx.x == 0) d.pending[slot] = AZ_PEND_NONE;
    __syncthreads();
    AZ_TOCK(d, g, pend == AZ_PEND_ROOT ? AZ_GC_T_ROOT : AZ_GC_T_BACKUP, tk_b);
}

// ------------------------------------------------------------- select phase
// One descent into evaluation slot k.  Leaves s_cells at the root position.
// returns 0 when the slot was filled, 1 when the round's descents must stop
// (fresh root, error, or the descent met a stub that an earlier slot of this
// round is already waiting on: the reference thread would block on that
// node's lock, mcts.hpp:224 -- here the descent is undone exactly).
__device__ int az_select(const AzTreeDev& d, int g, int k, uint8_t* s_cells, int* s_old) {
    const int A = d.A, N = d.N;
    const int slot = g * d.L + k;
    const int rt = uni(d.to_move[g]);
    const int hl = uni(d.hist_len[g]);
    if (uni(d.root_hist_len[g]) < 0) {                                  // fresh root: evaluate the root board
        az_write_planes(d, slot, s_cells, rt);
        if (threadIdx.x == 0) { d.pending[slot] = AZ_PEND_ROOT; d.path_len[slot] = 0; d.g_ctr[(size_t)g * AZ_GC_STRIDE + AZ_GC_EVALS] += 1ull; }
        return 1;
    }
    // per-ply records of this descent, all in LDS: while descending the wave issues loads only;
    // the virtual-loss updates and the path go to HBM in one parallel batch at the end
    int* s_old_n = s_old;
    const int pm = az_smem_plies(A);
    float* s_old_v = (float*)(s_old + pm);
    int* s_act = s_old + 2 * pm;
    int* s_path = s_old + 3 * pm;
    const int ar = uni(d.arena[g]);
    AzEdge* E = d.edges[ar] + (size_t)g * d.edge_cap;
    const uint16_t* EA = d.eact[ar] + (size_t)g * d.edge_cap;
    int* path = d.path + (size_t)slot * d.path_cap;
    const float vl = d.vl;
    const float cpf = (float)d.c_puct_i;

    // a node is its block of child edges: the root's starts at edge 0, a child's at edge.child; an
    // expanded node has one child per empty cell, so the block size needs no load either
    int fe = 0, ne = A - hl, par_n = uni(d.root_n[g]), player = rt, depth = 0, pend = AZ_PEND_NONE;
    int scanned = 0;                                                    // child edges read by this descent
    float t0 = 0.f, t1 = 0.f;
    TTS(10);
    const unsigned long long tk_s = AZ_TICK(d);
    unsigned long long tk_e = 0ull;
    for (;; --ne) {
        const double s = sqrt((double)(par_n - 1));                     // node.hpp:54 sqrt(int) -> double
        float best_val = -100.f;                                        // mcts.hpp:355
        int best_k = -1;
        AzEdge be = {0.f, 0, 0.f, -1};                                  // this lane's best candidate, whole
        int bact = 0;
        // four children per lane per trip (a 15x15 node has at most 225 = 3.5 x 64): their loads
        // are in flight together instead of one L2/HBM round trip per 64 children
        for (int k0 = threadIdx.x; k0 < ne; k0 += 4 * AZ_WAVE) {
            AzEdge ce[4];
            int ca[4];
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                const int k = min(k0 + u4 * AZ_WAVE, ne - 1);           // clamped: the duplicate never wins (strict >)
                ce[u4] = E[fe + k];
                ca[u4] = EA[fe + k];
            }
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                const int k = k0 + u4 * AZ_WAVE;
                const AzEdge e = ce[u4];
                const float cp = cpf * e.p;                             // int * float -> float
                float val;
                if (__ballot(e.n != 0) == 0ull) {
                    // 64 never-visited children (the common case): q = 0 and the division is by 1.0,
                    // so q + cp*s/(1+n) is cp*s to the bit -- skip the fp64 divide for the whole wave
                    val = (float)((double)cp * s);
                } else {
                    const float q = e.n != 0 ? __fdiv_rn(e.v, (float)e.n) : 0.f;
                    const double u = (double)cp * s / (double)(1 + e.n);
                    val = (float)((double)q + u);
                }
                if (k < ne && val > best_val) { best_val = val; best_k = k; be = e; bact = ca[u4]; }
            }
        }
        int win_k = best_k;
        for (int o = 32; o > 0; o >>= 1) {                              // first maximum in child order
            const float ov = __shfl_xor(best_val, o, AZ_WAVE);
            const int ok = __shfl_xor(win_k, o, AZ_WAVE);
            if (ok >= 0 && (win_k < 0 || ov > best_val || (ov == best_val && ok < win_k))) { best_val = ov; win_k = ok; }
        }
        win_k = uni(win_k);
        if (win_k < 0) {                                                // mcts.hpp:374-375 would throw
            if (threadIdx.x == 0) { d.err[g] = 2; d.status[g] = AZ_ST_IDLE; }
            return 1;
        }
        // the winner's edge comes from the lane that scanned it, not from memory again: lane
        // k % 64 scanned child k, and the wave's first maximum is also that lane's own first maximum
        const int owner = win_k & (AZ_WAVE - 1);
        const int ei = fe + win_k;
        AzEdge e;
        e.p = __shfl(be.p, owner, AZ_WAVE); e.n = __shfl(be.n, owner, AZ_WAVE);
        e.v = __shfl(be.v, owner, AZ_WAVE); e.child = __shfl(be.child, owner, AZ_WAVE);
        const int action = __shfl(bact, owner, AZ_WAVE);
        if (threadIdx.x == 0) {
            s_old_n[depth] = e.n; s_old_v[depth] = e.v; s_act[depth] = action; s_path[depth] = ei;
            s_cells[action] = (uint8_t)(player + 1);
        }
        e.n = (int)((float)e.n + vl);                                   // mcts.hpp:225-227 (stored below)
        e.v = e.v - vl;
        // the block is ONE wave: its LDS traffic is in order, no barrier needed to go one level deeper
        __builtin_amdgcn_wave_barrier();
        const int mover = player;
        player ^= 1;
        depth++;
        scanned += ne;
        const int child = uni(e.child);
        if (depth <= 8) TTS(10 + depth);
        if (child >= 0) { fe = child; par_n = uni(e.n); continue; }
        __syncthreads();
        if (child == AZ_CHILD_WAITING) {                                // nothing was stored yet: only the board to undo
            for (int t = threadIdx.x; t < depth; t += AZ_WAVE) s_cells[s_act[t]] = 0;
            __syncthreads();
            return 1;
        }
        // the descent stands: virtual loss n += vl (int += float), v -= vl on every edge of the path
        for (int t = threadIdx.x; t < depth; t += AZ_WAVE) {
            const int pe = s_path[t];
            E[pe].n = (int)((float)s_old_n[t] + vl);
            E[pe].v = s_old_v[t] - vl;
            path[t] = pe;
        }
        TTS(20);
#ifdef AZ_TREE_TIMING
        if (threadIdx.x == 0 && blockIdx.x == 7) az_tree_ts[22] = depth;
#endif
        AZ_TOCK(d, g, AZ_GC_T_SELECT, tk_s);                            // the descent (mcts.hpp:211-237)
        tk_e = AZ_TICK(d);
        if (child <= -2) {                                              // stored terminal reward, mcts.hpp:232-233
            pend = AZ_PEND_TERMINAL;
            t0 = child == -2 ? 1.f : (child == -3 ? -1.f : 0.f);
            t1 = 0.f - t0;
            break;
        }
        // stub: Env::step on the parent's state (mcts.hpp:261)
        const bool win = az_win_check(s_cells, N, d.K, action, mover + 1);
        const bool full = hl + depth == A;
        if (win || full) {                                              // terminal_add, mcts.hpp:266-269
            const int code = win ? (mover == 0 ? -2 : -3) : -4;
            if (threadIdx.x == 0) E[ei].child = code;
            pend = AZ_PEND_TERMINAL;
            t0 = win ? (mover == 0 ? 1.f : -1.f) : 0.f;
            t1 = 0.f - t0;
        } else {
            az_write_planes(d, slot, s_cells, player);
            if (threadIdx.x == 0) E[ei].child = AZ_CHILD_WAITING;
            pend = AZ_PEND_LEAF;
        }
        break;
    }
    if (threadIdx.x == 0) {
        d.pending[slot] = pend;
        d.path_len[slot] = depth;
        if (pend == AZ_PEND_TERMINAL) { d.pend_result[2 * slot] = t0; d.pend_result[2 * slot + 1] = t1; }
        unsigned long long* gc = d.g_ctr + (size_t)g * AZ_GC_STRIDE;    // per game: no hot shared counter in the round
        if (pend != AZ_PEND_TERMINAL) gc[AZ_GC_EVALS] += 1ull;
        gc[AZ_GC_DESCENTS] += 1ull; gc[AZ_GC_LEVELS] += (unsigned long long)depth; gc[AZ_GC_SCANNED] += (unsigned long long)scanned;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < depth; t += AZ_WAVE) s_cells[s_act[t]] = 0;
    __syncthreads();
    AZ_TOCK(d, g, AZ_GC_T_STEP, tk_e);                                  // Env::step + the board for the evaluator (mcts.hpp:248-271)
    if (d.timing && threadIdx.x == 0) d.g_ctr[(size_t)g * AZ_GC_STRIDE + AZ_GC_N_STEP] += 1ull;
    return 0;
}

// policy out = child n / root n (mcts.hpp:136-146)
__device__ void az_emit_policy(const AzTreeDev& d, int g) {
    const int A = d.A;
    float* out = d.out_policy + (size_t)g * A;
    for (int i = threadIdx.x; i < A; i += AZ_WAVE) out[i] = 0.f;
    __syncthreads();
    if (uni(d.root_hist_len[g]) < 0) return;
    const int ar = uni(d.arena[g]);
    const AzEdge* E = d.edges[ar] + (size_t)g * d.edge_cap;
    const uint16_t* EA = d.eact[ar] + (size_t)g * d.edge_cap;
    const int ne = A - uni(d.hist_len[g]);                              // the root's block: edges [0, ne)
    const float rn = (float)uni(d.root_n[g]);
    for (int k = threadIdx.x; k < ne; k += AZ_WAVE)
        out[EA[k]] = __fdiv_rn((float)E[k].n, rn);
    __syncthreads();
}

// -------------------------------------------------------- self-play control
// generate.cpp:84-118 for one slot after its search completed.
__device__ void az_selfplay_move(const AzTreeDev& d, int g, uint8_t* s_cells) {
    const int A = d.A, N = d.N;
    const int i = uni(d.move_idx[g]) + 1;                               // generate.cpp:82
    const int rt = uni(d.to_move[g]);
    // buffer.temporary_append(board, policy) (generate.cpp:85)
    uint8_t* rc = d.row_cells + ((size_t)g * A + (i - 1)) * A;
    float* rp = d.row_policy + ((size_t)g * A + (i - 1)) * A;
    const float* op = d.out_policy + (size_t)g * A;
    for (int c = threadIdx.x; c < A; c += AZ_WAVE) { rc[c] = s_cells[c]; rp[c] = op[c]; }
    if (threadIdx.x == 0) d.row_tomove[(size_t)g * A + (i - 1)] = (uint8_t)rt;

    const int ar = uni(d.arena[g]);
    const AzEdge* E = d.edges[ar] + (size_t)g * d.edge_cap;
    const uint16_t* EA = d.eact[ar] + (size_t)g * d.edge_cap;
    const int fe = 0, ne = A - uni(d.hist_len[g]);                      // the root's block
    int chosen = -1;
    if ((double)i < (double)d.avg_len[g] * 0.2) {                      // generate.cpp:88: sample ~ visits
        unsigned part = 0;
        for (int k = threadIdx.x; k < ne; k += AZ_WAVE) part += (unsigned)E[fe + k].n;
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, AZ_WAVE);
        const unsigned total = part;
        unsigned r[4];
        const unsigned ctr = d.sample_ctr[g];
        az_philox4(d.seed, (unsigned)(d.slot_base + g), ctr, r);
        const unsigned pick = (unsigned)(((unsigned long long)r[0] * total) >> 32);
        unsigned carry = 0;
        for (int c = 0; c < ne && chosen < 0; c += AZ_WAVE) {
            const int k = c + threadIdx.x;
            unsigned x = k < ne ? (unsigned)E[fe + k].n : 0u;
            for (int o = 1; o < AZ_WAVE; o <<= 1) {                     // inclusive scan
                const unsigned y = __shfl_up(x, o, AZ_WAVE);
                if ((int)threadIdx.x >= o) x += y;
            }
            const bool hit = k < ne && carry + x > pick;
            const unsigned long long m = __ballot(hit);
            if (m) chosen = c + (__ffsll((long long)m) - 1);
            carry += __shfl(x, AZ_WAVE - 1, AZ_WAVE);
        }
        if (threadIdx.x == 0) d.sample_ctr[g] = ctr + 1u;
    } else {                                                            // generate.cpp:95: first arg-max
        int bn = -1, bk = -1;
        for (int k = threadIdx.x; k < ne; k += AZ_WAVE) {
            const int n = E[fe + k].n;
            if (n > bn) { bn = n; bk = k; }
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int on = __shfl_xor(bn, o, AZ_WAVE), ok = __shfl_xor(bk, o, AZ_WAVE);
            if (ok >= 0 && (bk < 0 || on > bn || (on == bn && ok < bk))) { bn = on; bk = ok; }
        }
        chosen = bn > 0 ? bk : -1;
    }
    chosen = uni(chosen);
    // all-zero policy: torch::argmax returns 0 (cannot happen once root n > 1)
    const int action = chosen >= 0 ? EA[fe + chosen] : 0;
    const int child = chosen >= 0 ? uni(E[fe + chosen].child) : -1;
    const int child_n = chosen >= 0 ? uni(E[fe + chosen].n) : 0;

    // env.step on the real game (generate.cpp:103)
    const int hl = uni(d.hist_len[g]);
    if (threadIdx.x == 0) {
        s_cells[action] = (uint8_t)(rt + 1);
        d.cells[(size_t)g * A + action] = (uint8_t)(rt + 1);
        d.hist[(size_t)g * A + hl] = (uint16_t)action;
        d.hist_len[g] = hl + 1;
        d.to_move[g] = (uint8_t)(rt ^ 1);
        atomicAdd(&d.counters[2], 1ull);
    }
    __syncthreads();
    const bool win = az_win_check(s_cells, N, d.K, action, rt + 1);
    const bool done = win || hl + 1 == A;
    if (!done) {
        if (threadIdx.x == 0) { d.move_idx[g] = i; d.sims_done[g] = 0; d.sims_target[g] = d.budget; }
        __syncthreads();
        if (child >= 0) az_reroot(d, g, child, child_n); else az_drop_tree(d, g);
        return;
    }
    // game over: buffer.send_reward(reward); agent.clear(); stats (generate.cpp:110-118)
    const float r0 = win ? (rt == 0 ? 1.f : -1.f) : 0.f;
    int base = -1;
    if (threadIdx.x == 0) {
        unsigned long long old = atomicAdd(&d.ring_ctr[0], 0ull);
        for (;;) {
            if (old + (unsigned long long)i > (unsigned long long)d.ring_cap) { atomicAdd(&d.ring_ctr[1], (unsigned long long)i); break; }
            const unsigned long long prev = atomicCAS(&d.ring_ctr[0], old, old + (unsigned long long)i);
            if (prev == old) { base = (int)old; break; }
            old = prev;
        }
    }
    base = __shfl(base, 0, AZ_WAVE);
    if (base >= 0) {
        for (int r = 0; r < i; ++r) {
            const uint8_t* src_c = d.row_cells + ((size_t)g * A + r) * A;
            const float* src_p = d.row_policy + ((size_t)g * A + r) * A;
            uint8_t* dc = d.ring_cells + (size_t)(base + r) * A;
            float* dp = d.ring_policy + (size_t)(base + r) * A;
            for (int c = threadIdx.x; c < A; c += AZ_WAVE) { dc[c] = src_c[c]; dp[c] = src_p[c]; }
            if (threadIdx.x == 0) {
                d.ring_tomove[base + r] = d.row_tomove[(size_t)g * A + r];
                d.ring_reward[2 * (size_t)(base + r)] = r0;
                d.ring_reward[2 * (size_t)(base + r) + 1] = 0.f - r0;
            }
        }
    }
    if (threadIdx.x == 0) {
        const long long gc = d.game_count[g] + 1;                       // generate.cpp:25-29
        d.game_count[g] = gc;
        const float al = d.avg_len[g];
        d.avg_len[g] = al + ((float)i - al) / (float)gc;
        atomicAdd(&d.counters[3], 1ull);
        d.final_reward[2 * g] = r0; d.final_reward[2 * g + 1] = 0.f - r0;
        // env.reset(), then the warm-up search (generate.cpp:75-77): move_idx -1 marks it
        d.hist_len[g] = 0; d.to_move[g] = 0; d.move_idx[g] = -1;
        d.sims_done[g] = 0; d.sims_target[g] = d.budget;
    }
    for (int c = threadIdx.x; c < A; c += AZ_WAVE) { d.cells[(size_t)g * A + c] = 0; s_cells[c] = 0; }
    __syncthreads();
    az_drop_tree(d, g);
}

// ------------------------------------------------------------------ kernels
// One lockstep round.  phase_mask bit0: apply the pending evaluation,
// bit1: advance (finish search / play move / select the next leaf).
__device__ __forceinline__ void az_round_once(const AzTreeDev& d, int phase_mask, int mode, int g, unsigned char* smem) {
    uint8_t* s_cells = smem;
    int* s_old = (int*)(smem + az_smem_cells(d.A));
    TTS(0);
    if (uni(d.status[g]) != AZ_ST_RUN) return;
    TTS(1);
    for (int c = threadIdx.x; c < d.A; c += AZ_WAVE) s_cells[c] = d.cells[(size_t)g * d.A + c];
    __syncthreads();
    TTS(2);
    if (phase_mask & 1)
        for (int k = 0; k < d.L; ++k) {
            az_apply(d, g, k, s_cells, s_old + 2 * az_smem_plies(d.A));
            if (uni(d.status[g]) != AZ_ST_RUN) return;
        }
    TTS(3);
    if (!(phase_mask & 2)) return;
    if (uni(d.pending[g * d.L]) != AZ_PEND_NONE) return;                // select already issued
    TTS(4);
    if (uni(d.sims_done[g]) >= uni(d.sims_target[g])) {
        if (threadIdx.x == 0) atomicAdd(&d.counters[0], (unsigned long long)d.sims_done[g]);
        if (mode == AZ_MODE_SELFPLAY && uni(d.move_idx[g]) < 0) {
            // the warm-up search of a new game is over (generate.cpp:77); the first
            // move's own search continues on the same root
            __syncthreads();
            if (threadIdx.x == 0) { d.move_idx[g] = 0; d.sims_done[g] = 0; d.sims_target[g] = d.budget; }
            __syncthreads();
        } else {
            az_emit_policy(d, g);
            if (mode == AZ_MODE_SEARCH) {
                if (threadIdx.x == 0) d.status[g] = AZ_ST_IDLE;
                return;
            }
            az_selfplay_move(d, g, s_cells);
        }
    }
    // up to L descents, never more than the simulations still owed (mcts.hpp:161-174)
    const int owed = uni(d.sims_target[g]) - uni(d.sims_done[g]);
    TTS(5);
    for (int k = 0; k < d.L && k < owed; ++k)
        if (az_select(d, g, k, s_cells, s_old)) break;
    TTS(6);
}

// A game's round.  A descent that ends in a terminal leaf whose reward is already stored needs no evaluation
// (mcts.hpp:232-233: the reference returns the stored reward and goes straight to __backup), so with d.chain > 0 up
// to that many such simulations are applied and followed by the next select inside the same round, until a leaf
// that does need the evaluator has been issued.  The order of a tree's simulations is unchanged; only how many of
// them fit between two evaluations is.
__device__ __forceinline__ void az_round_body(const AzTreeDev& d, int phase_mask, int mode, int g, unsigned char* smem) {
    int chained = 0;
    for (;;) {
        az_round_once(d, phase_mask, mode, g, smem);
        if (phase_mask != 3 || d.L != 1 || chained >= d.chain) break;
        __syncthreads();
        if (uni(d.status[g]) != AZ_ST_RUN || uni(d.pending[g]) != AZ_PEND_TERMINAL) break;
        ++chained;
    }
}

// The per-game state table (some sixty pointers) is read from the kernel-argument segment where it is used, through
// a pointer the compiler cannot see through: kept in registers across the loops of a round it costs more scalar
// registers than a wave has (spilled to vector lanes, and those to memory in the 64-register pair kernel).
typedef const __attribute__((address_space(4))) AzTreeDev* AzTreeArgs;
#define AZ_TREE_ARGS(p_) ({ AzTreeArgs q_ = (p_); asm volatile("" : "+s"(q_)); (const AzTreeDev*)q_; })

extern "C" __global__ void __launch_bounds__(AZ_WAVE)
az_round_kernel(AzTreeDev, int phase_mask, int mode, int g0) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AzTreeDev* d = AZ_TREE_ARGS((AzTreeArgs)__builtin_amdgcn_kernarg_segment_ptr());
    az_round_body(*d, phase_mask, mode, g0 + blockIdx.x, smem);        // a launch covers games [g0, g0 + gridDim.x)
}

// The tree half of the persistent pair (AzPairDev, az_dev.h): a wave takes its games in turn, round after round;
// a game's step runs as soon as that game's evaluation is there, whatever the other games are doing.  Same
// az_round_body as the lockstep kernel, so a game's tree is the same bit for bit: games share no tree state.
extern "C" __global__ void __launch_bounds__(AZ_WAVE) __attribute__((amdgpu_waves_per_eu(8, 8)))
az_round_pair_kernel(AzTreeDev, AzPairDev pr, int mode) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const AzTreeArgs ka = (AzTreeArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const int G = ka->G;
    for (int i = 1; i <= pr.rounds; ++i) {
        const unsigned want = pr.base + (unsigned)i;
        for (int g = blockIdx.x; g < G; g += gridDim.x) {
            const AzTreeDev& d = *AZ_TREE_ARGS(ka);
            // the tree is never the hurry (its planes are wanted a whole tile later): poll seldom, stay out of the tower's way
            const bool ok = az_pair_wait<100>(pr.eval_seq + g, want, pr.abort, pr.timeout_ticks, 0x80000000u | (unsigned)g);
            if (__ballot(!ok) != 0ull) return;                          // a wait ran out somewhere: leave
            // the evaluation was stored write-through by another CU: drop this CU's L1 copies before reading it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            az_round_body(d, 3, mode, g, smem);
            __syncthreads();
            // The leaf's planes go to the tower's CU as whole words stored write-through, which need no write-back of
            // this XCD's L2 (a release fence would flush every tree wave's dirty edges, in the way of the towers'
            // weight stream): read back what this wave wrote (it sees its own stores), pack, store, drain, flag.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            {
                const uint8_t* src = d.leaf_planes + (size_t)g * 3 * d.A;
                unsigned* dst = pr.planes_wt + (size_t)g * (pr.plane_stride >> 2);
                const int nw = (3 * d.A + 3) >> 2;
                for (int w = threadIdx.x; w < nw; w += AZ_WAVE) {
                    unsigned v = 0;
                    for (int k = 0; k < 4; ++k)
                        if (4 * w + k < 3 * d.A) v |= (unsigned)src[4 * w + k] << (8 * k);
                    __hip_atomic_store(dst + w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (threadIdx.x == 0) {
                __hip_atomic_store(pr.plane_seq + g, want + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pr.busy[g] += __builtin_amdgcn_s_memrealtime() - t0;
            }
        }
    }
}

// search_probs entry: root reuse decision (mcts.hpp:121) for every game.
extern "C" __global__ void __launch_bounds__(AZ_WAVE)
az_begin_kernel(AzTreeDev d, int budget) {
    const int g = blockIdx.x;
    const unsigned long long tk_r = AZ_TICK(d);
    const int ok = az_make_root(d, g);
    AZ_TOCK(d, g, AZ_GC_T_ROOT, tk_r);                                  // find + prune (mcts.hpp:385-421)
    if (threadIdx.x == 0) {
        d.sims_done[g] = 0;
        // the loop body runs once even for budget <= 0 (mcts.hpp:161-174)
        d.sims_target[g] = budget > 1 ? budget : 1;
        for (int k = 0; k < d.L; ++k) d.pending[g * d.L + k] = AZ_PEND_NONE;
        d.status[g] = ok ? AZ_ST_RUN : AZ_ST_IDLE;
    }
    if (!ok) {
        float* out = d.out_policy + (size_t)g * d.A;
        for (int i = threadIdx.x; i < d.A; i += AZ_WAVE) out[i] = 0.f;
    }
}

extern "C" __global__ void az_selfplay_begin_kernel(AzTreeDev d) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= d.G) return;
    d.status[g] = AZ_ST_RUN;
    for (int k = 0; k < d.L; ++k) d.pending[g * d.L + k] = AZ_PEND_NONE;
    d.sims_done[g] = 0;
    d.sims_target[g] = d.budget;
    d.move_idx[g] = -1;                                                 // warm-up search first (generate.cpp:77)
    d.avg_len[g] = 25.f;                                                // generate.cpp:69
    d.game_count[g] = 0;
    d.sample_ctr[g] = 0;
    d.root_hist_len[g] = -1; d.n_nodes[g] = 0; d.n_edges[g] = 0; d.root_n[g] = 0;
    d.hist_len[g] = 0; d.to_move[g] = 0; d.game_over[g] = 0;
    for (int c = 0; c < d.A; ++c) d.cells[(size_t)g * d.A + c] = 0;
}

// Env::step for the real games (generate.cpp:103), batched.
extern "C" __global__ void __launch_bounds__(AZ_WAVE)
az_env_step_kernel(AzTreeDev d, const int* actions, float* reward, uint8_t* done, int* illegal) {
    __shared__ uint8_t s_cells[AZ_CELLS_MAX];
    const int g = blockIdx.x;
    const int A = d.A;
    const int action = uni(actions[g]);
    if (threadIdx.x == 0) { reward[2 * g] = 0.f; reward[2 * g + 1] = 0.f; done[g] = d.game_over[g]; }
    if (action < 0) return;
    for (int c = threadIdx.x; c < A; c += AZ_WAVE) s_cells[c] = d.cells[(size_t)g * A + c];
    __syncthreads();
    if (action >= A || s_cells[action] != 0 || uni(d.game_over[g])) {
        if (threadIdx.x == 0) atomicAdd(illegal, 1);
        return;
    }
    const int mover = uni(d.to_move[g]);
    const int hl = uni(d.hist_len[g]);
    if (threadIdx.x == 0) {
        s_cells[action] = (uint8_t)(mover + 1);
        d.cells[(size_t)g * A + action] = (uint8_t)(mover + 1);
        d.hist[(size_t)g * A + hl] = (uint16_t)action;
        d.hist_len[g] = hl + 1;
        d.to_move[g] = (uint8_t)(mover ^ 1);
    }
    __syncthreads();
    const bool win = az_win_check(s_cells, d.N, d.K, action, mover + 1);
    const bool over = win || hl + 1 == A;
    if (threadIdx.x == 0) {
        const float r0 = win ? (mover == 0 ? 1.f : -1.f) : 0.f;
        reward[2 * g] = r0; reward[2 * g + 1] = 0.f - r0;
        done[g] = over;
        d.game_over[g] = over;
        d.final_reward[2 * g] = r0; d.final_reward[2 * g + 1] = 0.f - r0;
    }
}

extern "C" __global__ void az_env_reset_kernel(AzTreeDev d) {
    const int g = blockIdx.x;
    for (int c = threadIdx.x; c < d.A; c += blockDim.x) d.cells[(size_t)g * d.A + c] = 0;
    if (threadIdx.x == 0) { d.to_move[g] = 0; d.hist_len[g] = 0; d.game_over[g] = 0; }
}

// Env::get_board for the real games: u8 planes [G,3,N,N]; legal mask [G,N,N].
extern "C" __global__ void az_env_export_kernel(AzTreeDev d, uint8_t* planes, uint8_t* mask) {
    const int g = blockIdx.x;
    const int A = d.A;
    const int tm = d.to_move[g];
    for (int c = threadIdx.x; c < A; c += blockDim.x) {
        const uint8_t v = d.cells[(size_t)g * A + c];
        if (planes) {
            planes[(size_t)g * 3 * A + c] = v == 1;
            planes[(size_t)g * 3 * A + A + c] = v == 2;
            planes[(size_t)g * 3 * A + 2 * A + c] = (uint8_t)tm;
        }
        if (mask) mask[(size_t)g * A + c] = v == 0 && !d.game_over[g];
    }
}

// Integer-hash pseudo network (parity evaluator): one wave per board.
extern "C" __global__ void __launch_bounds__(AZ_WAVE)
az_eval_hash_kernel(const uint8_t* planes, float* policy, float* value, int A) {
    const int g = blockIdx.x;
    const uint8_t* pl = planes + (size_t)g * 3 * A;
    unsigned long long H = 0;
    for (int i = threadIdx.x; i < 3 * A; i += AZ_WAVE)
        H ^= az_splitmix64((unsigned long long)i * 256ull + pl[i]);
    for (int o = 32; o > 0; o >>= 1) H ^= __shfl_xor(H, o, AZ_WAVE);
    unsigned sum = 0;
    for (int i = threadIdx.x; i < A; i += AZ_WAVE) {
        const unsigned long long r = (az_splitmix64(H + (unsigned long long)i + 1ull) >> 40) & 0xFFFFull;
        sum += 1u + (unsigned)((r * r * r) >> 32);
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, AZ_WAVE);
    const float fs = (float)sum;
    for (int i = threadIdx.x; i < A; i += AZ_WAVE) {
        const unsigned long long r = (az_splitmix64(H + (unsigned long long)i + 1ull) >> 40) & 0xFFFFull;
        const unsigned w = 1u + (unsigned)((r * r * r) >> 32);
        policy[(size_t)g * A + i] = __fdiv_rn((float)w, fs);
    }
    if (threadIdx.x == 0) {
        const int t = (int)((az_splitmix64(H ^ 0xABCDEFull) >> 40) & 0xFFFFull);
        const float v0 = __fdiv_rn((float)t - 32768.0f, 40000.0f);
        value[2 * g] = v0;
        value[2 * g + 1] = -v0;
    }
}

// ------------------------------------------------------------ host launchers
#include "az_tree.h"

static inline size_t tree_smem(const AzTreeDev& d) { return (size_t)az_smem_bytes(d.A); }
static_assert(16 * ((21 * 21 + 15) / 16) + 4 * 448 * 4 <= AZ_TREE_SMEM_BYTES, "az_tree.h states the tree block's largest LDS");

extern "C" int az_tree_configure(const AzTreeDev*) { return 0; }
extern "C" void az_launch_round(hipStream_t s, const AzTreeDev* d, int phase_mask, int mode, int g0, int count) {
    hipLaunchKernelGGL(az_round_kernel, dim3(count), dim3(AZ_WAVE), tree_smem(*d), s, *d, phase_mask, mode, g0);
}
// One wave in front of the tree kernel on its stream: ends when every tower workgroup of the launch has begun (or the
// wait runs out: the tree waves then find the abort word raised and leave at once).
extern "C" __global__ void __launch_bounds__(AZ_WAVE) az_pair_gate_kernel(AzPairDev pr) {
    if (threadIdx.x == 0) (void)az_pair_wait<20>(pr.start + 1, pr.started_want, pr.abort, pr.timeout_ticks, 0xC0000000u);
}
extern "C" void az_launch_round_pair(hipStream_t s, const AzTreeDev* d, const AzPairDev* pr, int mode, int blocks) {
    hipLaunchKernelGGL(az_pair_gate_kernel, dim3(1), dim3(AZ_WAVE), 0, s, *pr);
    hipLaunchKernelGGL(az_round_pair_kernel, dim3(blocks), dim3(AZ_WAVE), tree_smem(*d), s, *d, *pr, mode);
}
extern "C" void az_launch_begin(hipStream_t s, const AzTreeDev* d, int budget) {
    hipLaunchKernelGGL(az_begin_kernel, dim3(d->G), dim3(AZ_WAVE), tree_smem(*d), s, *d, budget);
}
// The tallies az_get_stats reports, reduced over the games on the device: out[0..15] = the context's counters,
// out[16..] = evals, descents, tree levels, children scanned, children expanded, restarts, simulations of searches
// in flight, games in error (sums), then the largest node and edge counts (maxima).
extern "C" __global__ void __launch_bounds__(256) az_stats_kernel(AzTreeDev d, unsigned long long* out) {
    __shared__ unsigned long long s_sum[8][4];
    __shared__ int s_max[2][4];
    unsigned long long sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tks[6] = {0, 0, 0, 0, 0, 0};                      // phase clocks, see az_tree_timing
    int mx[2] = {0, 0};
    for (int g = threadIdx.x; g < d.G; g += 256) {
        const unsigned long long* q = d.g_ctr + (size_t)g * AZ_GC_STRIDE;
        sum[0] += q[AZ_GC_EVALS]; sum[1] += q[AZ_GC_DESCENTS]; sum[2] += q[AZ_GC_LEVELS];
        sum[3] += q[AZ_GC_SCANNED]; sum[4] += q[AZ_GC_EXPANDED]; sum[5] += q[AZ_GC_RESTARTS];
        for (int k = 0; k < 6; ++k) tks[k] += q[AZ_GC_T_SELECT + k];
        if (d.status[g] == AZ_ST_RUN) sum[6] += (unsigned long long)d.sims_done[g];
        sum[7] += d.err[g] != 0;
        mx[0] = max(mx[0], max(d.g_hw_nodes[g], d.n_nodes[g]));
        mx[1] = max(mx[1], max(d.g_hw_edges[g], d.n_edges[g]));
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int k = 0; k < 8; ++k) {
        for (int o = 32; o > 0; o >>= 1) sum[k] += __shfl_xor(sum[k], o, 64);
        if (lane == 0) s_sum[k][wv] = sum[k];
    }
    for (int k = 0; k < 2; ++k) {
        for (int o = 32; o > 0; o >>= 1) mx[k] = max(mx[k], __shfl_xor(mx[k], o, 64));
        if (lane == 0) s_max[k][wv] = mx[k];
    }
    for (int k = 0; k < 6; ++k) {                                        // rare path: plain atomics into the output words
        for (int o = 32; o > 0; o >>= 1) tks[k] += __shfl_xor(tks[k], o, 64);
        if (lane == 0 && tks[k]) atomicAdd(&out[26 + k], tks[k]);
    }
    __syncthreads();
    if (threadIdx.x < 16) out[threadIdx.x] = d.counters[threadIdx.x];
    if (threadIdx.x < 8) out[16 + threadIdx.x] = s_sum[threadIdx.x][0] + s_sum[threadIdx.x][1] + s_sum[threadIdx.x][2] + s_sum[threadIdx.x][3];
    if (threadIdx.x < 2)
        out[24 + threadIdx.x] = (unsigned long long)max(max(s_max[threadIdx.x][0], s_max[threadIdx.x][1]), max(s_max[threadIdx.x][2], s_max[threadIdx.x][3]));
}
extern "C" void az_launch_stats(hipStream_t s, const AzTreeDev* d, unsigned long long* out) {
    hipLaunchKernelGGL(az_stats_kernel, dim3(1), dim3(256), 0, s, *d, out);
}

extern "C" void az_launch_selfplay_begin(hipStream_t s, const AzTreeDev* d) {
    hipLaunchKernelGGL(az_selfplay_begin_kernel, dim3((d->G + 63) / 64), dim3(64), 0, s, *d);
}
extern "C" void az_launch_env_step(hipStream_t s, const AzTreeDev* d, const int* actions, float* reward,
                                   uint8_t* done, int* illegal) {
    hipLaunchKernelGGL(az_env_step_kernel, dim3(d->G), dim3(AZ_WAVE), 0, s, *d, actions, reward, done, illegal);
}
extern "C" void az_launch_env_reset(hipStream_t s, const AzTreeDev* d) {
    hipLaunchKernelGGL(az_env_reset_kernel, dim3(d->G), dim3(64), 0, s, *d);
}
extern "C" void az_launch_env_export(hipStream_t s, const AzTreeDev* d, uint8_t* planes, uint8_t* mask) {
    hipLaunchKernelGGL(az_env_export_kernel, dim3(d->G), dim3(64), 0, s, *d, planes, mask);
}
extern "C" void az_launch_eval_hash(hipStream_t s, const uint8_t* planes, float* policy, float* value, int B, int A) {
    hipLaunchKernelGGL(az_eval_hash_kernel, dim3(B), dim3(AZ_WAVE), 0, s, planes, policy, value, A);
}
