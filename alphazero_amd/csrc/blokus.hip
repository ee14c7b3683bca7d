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
    int*      legal_count;  // [G] legal actions of the player to move (0 once over)
    uint32_t* legal_bits;   // [G][91][20] row masks
    float*    final_reward; // [G][4]
};

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

// Places `action` (known legal) for the player to move, then hands the turn to the next player who can move,
// or ends the game.  s_own / s_used / s_last / s_meta {to_move, over, pieces placed, legal count} hold the
// game (and stay current, so a kernel can play on without re-reading HBM); the state arrays are updated too.
__device__ void azb_play(const AzbDev& d, int g, int action, uint32_t* s_own, uint2* s_rows, const uint2* s_tab,
                         uint32_t* s_bits, int* s_cnt, uint32_t* s_used, int* s_last, int* s_meta, float* reward, uint8_t* done) {
    const int lane = threadIdx.x;
    const int p = s_meta[0];
    const int o = action / AZB_CELLS, y0 = action % AZB_CELLS / AZB_N, x0 = action % AZB_N;
    __syncthreads();
    if (lane < c_orient[o].h) s_own[p * AZB_N + y0 + lane] |= (uint32_t)c_orient[o].rows[lane] << x0;
    if (lane == 0) { s_used[p] |= 1u << c_orient[o].piece; s_last[p] = c_orient[o].piece; }
    __syncthreads();
    int next = -1, cnt = 0;
    for (int k = 1; k <= AZB_PLAYERS; ++k) {
        const int q = (p + k) & 3;
        cnt = azb_scan(s_own, s_used[q], q, s_rows, s_tab, s_bits, s_cnt);
        if (cnt > 0) { next = q; break; }
    }
    // write the position back: only player p's rows and masks changed
    if (lane < AZB_N) d.own[((size_t)g * 4 + p) * AZB_N + lane] = s_own[p * AZB_N + lane];
    for (int i = lane; i < AZB_SET_WORDS; i += 64) d.legal_bits[(size_t)g * AZB_SET_WORDS + i] = next >= 0 ? s_bits[i] : 0u;
    if (lane == 0) {
        d.used[g * 4 + p] = s_used[p];
        d.last_piece[g * 4 + p] = (uint8_t)s_last[p];
        s_meta[2] += 1; d.moves[g] = s_meta[2];
        s_meta[3] = next >= 0 ? cnt : 0; d.legal_count[g] = s_meta[3];
        float r[4] = {0.f, 0.f, 0.f, 0.f};
        if (next >= 0) { s_meta[0] = next; d.to_move[g] = (uint8_t)next; }
        else {                                                       // nobody can move: score the hands
            s_meta[1] = 1; d.over[g] = 1;
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
extern "C" __global__ void __launch_bounds__(64) azb_reset_kernel(AzbDev d) {
    AZB_SMEM_DECL;
    (void)s_last; (void)s_meta;
    const int g = blockIdx.x, lane = threadIdx.x;
    for (int i = lane; i < 4 * AZB_N; i += 64) { s_own[i] = 0u; d.own[(size_t)g * 4 * AZB_N + i] = 0u; }
    if (lane < 4) { s_used[lane] = 0u; d.used[g * 4 + lane] = 0u; d.last_piece[g * 4 + lane] = 0xFF; d.final_reward[g * 4 + lane] = 0.f; }
    __syncthreads();
    azb_load_table(s_tab);
    const int cnt = azb_scan(s_own, 0u, 0, s_rows, s_tab, s_bits, s_cnt);
    for (int i = lane; i < AZB_SET_WORDS; i += 64) d.legal_bits[(size_t)g * AZB_SET_WORDS + i] = s_bits[i];
    if (lane == 0) { d.to_move[g] = 0; d.over[g] = 0; d.moves[g] = 0; d.legal_count[g] = cnt; }
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

// ------------------------------------------------------------------------------------------------ host
struct az_blokus {
    int device = 0;
    hipStream_t stream = nullptr;
    AzbDev d;
    std::vector<void*> allocs;
    int* d_actions = nullptr; float* d_reward = nullptr; uint8_t* d_done = nullptr; int* d_illegal = nullptr;
    uint8_t* d_export = nullptr;      // [G][91*400] mask or [G][5][400] planes
};

static int bfail(int code, const std::string& m) { return az_internal_fail(code, m.c_str()); }
#define BHIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return bfail(AZ_E_HIP, std::string(#call " failed: ") + hipGetErrorString(e_)); } while (0)
#define BNEED(p) do { if (!(p)) return bfail(AZ_E_ARG, "null argument: " #p); } while (0)

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
    b->d.G = G; b->d.seed = seed; b->d.slot_base = game_slot_base;
    int rc = AZ_OK;
#define BA(ptr, n) do { if (rc == AZ_OK) rc = balloc(b, &(ptr), (size_t)(n)); } while (0)
    BA(b->d.own, (size_t)G * 4 * AZB_N); BA(b->d.used, (size_t)G * 4); BA(b->d.last_piece, (size_t)G * 4);
    BA(b->d.to_move, G); BA(b->d.over, G); BA(b->d.moves, G); BA(b->d.legal_count, G);
    BA(b->d.legal_bits, (size_t)G * AZB_SET_WORDS); BA(b->d.final_reward, (size_t)G * 4);
    BA(b->d_actions, G); BA(b->d_reward, (size_t)G * 4); BA(b->d_done, G); BA(b->d_illegal, 4);
    BA(b->d_export, (size_t)G * AZB_ACTIONS);
#undef BA
    if (rc == AZ_OK && hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) rc = bfail(AZ_E_HIP, "hipStreamCreate failed");
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
