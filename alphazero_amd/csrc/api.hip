// api.hip -- the C ABI of libazhip.so (include/az_hip.h): context, weights,
// batched env, tower forward, lockstep search and device-side self-play.
// Host code only; kernels live in tree.hip and tower.hip.
#include "../../include/az_hip.h"
#include "az_dev.h"
#include "az_rules.h"
#include "az_tower.h"
#include "az_tree.h"
#include "az_weights.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

// Activation images of az_tower_forward_static: one array per GPU, alive until the process ends.  In host
// code the array's name is only a handle for hipGetSymbolAddress; every use of the memory goes through
// the per-device address kept in g_scratch_ptr.
#define AZ_STATIC_SCRATCH_BYTES (64u << 20)
#define AZ_MAX_DEVICES 64
static __device__ __align__(256) unsigned char scratch[AZ_STATIC_SCRATCH_BYTES];
static unsigned char* g_scratch_ptr[AZ_MAX_DEVICES];
static hipStream_t g_scratch_user[AZ_MAX_DEVICES];        // the stream of the last az_tower_forward_static call on the device
static std::mutex g_scratch_mu;
// Pair launches (az_selfplay_run_pair) of different contexts on one GPU must not overlap: each wants every CU for its
// tower workgroups and the one tree-wave slot beside each, and two half-resident pairs would wait for each other
// until their time-outs.  One event per GPU, recorded behind every pair launch; the next one waits for it.
// Streams of contexts that are gone, per GPU, for the contexts to come (first = a context's stream, second = the stream of
// its pair's tree kernel, null when it had none).  A process that makes and drops contexts all day keeps its hardware
// queues this way: every new queue is a round of taking all queues of the process off the GPU and putting them back,
// and a queue's scratch memory is set up anew in the middle of its first kernel that needs some.
static std::vector<std::pair<hipStream_t, hipStream_t>> g_stream_pool[AZ_MAX_DEVICES];
static std::mutex g_stream_pool_mu;
static hipEvent_t g_pair_done[AZ_MAX_DEVICES];          // per GPU: the end of the last pair launch of any context
static const void* g_pair_last[AZ_MAX_DEVICES];         // ... and whose it was
static std::mutex g_pair_mu[AZ_MAX_DEVICES];

static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    return fail(AZ_E_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); } while (0)
#define AZCHK(call) do { int r_ = (call); if (r_ != AZ_OK) return r_; } while (0)
#define NEED(ptr) do { if (!(ptr)) return fail(AZ_E_ARG, "null argument: %s", #ptr); } while (0)
// entry points that touch per-game state: a tower-only context (see ctx_build) has none
#define NEED_GAMES(c) do { if ((c)->tower_only) return fail(AZ_E_STATE, "this context is a tower only (c_in %d, c_out %d, " \
    "value_dim %d): it holds no games and does not fit the m,n,k search", (c)->c_in, (c)->c_out, (c)->vd); } while (0)

extern "C" const char* az_last_error(void) { return g_err.c_str(); }
// for the library's other translation units (group.hip): record an error for az_last_error
extern "C" int az_internal_fail(int code, const char* msg) { g_err = msg ? msg : ""; return code; }

struct az_ctx {
    az_config cfg;
    hipStream_t stream = nullptr;
    AzTreeDev d;
    std::vector<void*> allocs;
    int A = 0, P = 0, W = 0, Gd = 0, C = 0, C_pad = 0, num_res = 0;
    int c_in = 3, c_out = 1, vd = 2;                      // tower input planes, policy channels, value entries
    int Bmax = 0;
    // tower
    bool have_weights = false;
    unsigned short *actX = nullptr, *actH = nullptr;      // per-layer path: [rows][C_pad] 16-bit inputs of the convolutions
    float* actR = nullptr;                                // per-layer path: [rows][C_pad] fp32 residual stream
    unsigned short* actL = nullptr;                       // per-layer path, default kernel: the stream as actX + actL (16-bit remainder);
                                                          // actR then holds only the last block's output, for the heads
    bool hilo = false;
    size_t act_rows = 0;
    float *stem_w = nullptr, *stem_b = nullptr;           // [C_pad][27], [C_pad]
    unsigned short* stem_wbf = nullptr;                   // [128][32] bf16, fused path
    unsigned short* conv_wf = nullptr;                    // [layer][tap][n=128][k=128] bf16, fused path
    bool fused = false;                                   // whole tower in one launch
    int conv_kind = 0;                                    // per-layer convolution kernel (az_conv_kind), fixed at creation
    int prec = 0;                                         // 0 bf16, 1 fp16 operands (kernel template index) in use now
    bool split = false;                                   // AZ_PREC_FP16X3: operands as hi + lo pairs, convolutions over 3x the channels
    bool tower_only = false;                              // a tower whose shape no m,n,k search can use: no tree state allocated
    unsigned short *catX = nullptr, *catH = nullptr;      // split: [rows][3 C_pad] = [hi | lo | hi] of the stream / the hidden activation
    float* actS = nullptr;                                // split, AZ_X3_SEPARATE_SPLIT only: fp32 image of a block's hidden activation
    unsigned short* conv_w3 = nullptr;                    // split: [W_hi | W_hi | W_lo] images, 3x conv_w
    int* d_range = nullptr;                               // [4] range tally of the fold's dry run (weights.hip)
    unsigned long long* d_sat = nullptr;                  // [1] az_stats.saturated
    az_weights_info winfo = {};
    unsigned short* conv_w = nullptr;                     // packed chunks, 2*num_res layers
    float* conv_b = nullptr;                              // [2*num_res][C_pad]
    float* head_p = nullptr;
    float* head_pt = nullptr;                             // fused path: 1x1 weights as {policy, value} pairs, fc1 transposed [A][64]
    size_t conv_layer_elems = 0;
    float* blob_stage = nullptr;                          // the fp32 blob as loaded, input of the fold kernels
    uint8_t* tower_in = nullptr;                          // staging for az_tower_forward [Bmax][3][A]
    float *tower_pol = nullptr, *tower_val = nullptr;
    // misc device scratch
    int* d_actions = nullptr; float* d_reward = nullptr; uint8_t* d_done = nullptr; int* d_illegal = nullptr;
    uint8_t* d_export = nullptr;                          // [G][3][A] planes or [G][A] mask of az_env_boards / az_env_legal_mask
    unsigned long long* d_stats = nullptr;                // [32] az_get_stats' reduction
    // the two replay rings: launches append to rings[active] (= d.ring_*); az_replay_flip retires it
    struct Ring { uint8_t* cells; uint8_t* tomove; float* policy; float* reward; } rings[2] = {};
    int active = 0;
    hipEvent_t flip_ev = nullptr;                         // recorded at the flip: every writer of the retired ring is in front of it
    int mode = AZ_MODE_SEARCH;
    bool search_active = false;
    int64_t rounds = 0;
    // the persistent pair (az_selfplay_run_pair): the tree kernel's own stream, the events that order it against the
    // context's stream, the hand-over words, what the launches so far have published
    hipStream_t stream2 = nullptr;
    hipStream_t pooled2 = nullptr;                      // (while the context is built: a second stream taken over from a context before it)
    hipEvent_t pair_join = nullptr;                     // the tree kernel of the last pair launch has ended
    unsigned pair_started = 0;                          // tower workgroups launched by the pair launches so far
    int pair_queue_attempts = 0;                        // streams tried until one ran side by side with the context's (0: none did)
    bool other_work = false;                            // tower launches / lockstep rounds queued on the stream since the last pair launch
    AzPairDev pair = {};
    unsigned pair_base = 0;
    int n_cu = 0;
    int64_t pair_launches = 0, pair_rounds = 0;
    unsigned long long tree_ticks[6] = {0, 0, 0, 0, 0, 0};   // phase clocks summed over the games at the last az_get_stats
    std::vector<std::pair<size_t, int>> pair_spans;       // (first event index, rounds) of the sampled pair launches
    // HIP-event sampling of the conv / tree kernels on the context's stream
    int prof_every = 0;
    int64_t prof_calls = 0;
    std::vector<hipEvent_t> prof_pool;
    size_t prof_used = 0;
    std::vector<std::pair<size_t, int>> prof_spans;       // (first event index, kind 0 conv / 1 tree / 2 empty)
    double prof_empty_us = 0.0;                           // mean empty span of the last az_profile_read
    int64_t prof_empty_n = 0;
};

static bool prof_begin(az_ctx* c, int kind, bool sampled, hipStream_t s) {
    if (!sampled || c->prof_used + 2 > c->prof_pool.size()) return false;
    c->prof_spans.push_back({c->prof_used, kind});
    (void)hipEventRecord(c->prof_pool[c->prof_used], s);
    return true;
}
static void prof_end(az_ctx* c, bool on, hipStream_t s) {
    if (!on) return;
    (void)hipEventRecord(c->prof_pool[c->prof_used + 1], s);
    c->prof_used += 2;
}
static int sync_all(az_ctx* c) {
    HIPCHK(hipStreamSynchronize(c->stream));
    return AZ_OK;
}

template <class T>
static int dalloc(az_ctx* c, T** p, size_t n, bool zero = true) {
    void* q = nullptr;
    const size_t bytes = std::max<size_t>(n * sizeof(T), 16);
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) return fail(AZ_E_HIP, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    if (zero) { e = hipMemset(q, 0, bytes); if (e != hipSuccess) return fail(AZ_E_HIP, "hipMemset failed: %s", hipGetErrorString(e)); }
    c->allocs.push_back(q);
    *p = (T*)q;
    return AZ_OK;
}

static size_t weights_size(int N, int c_in, int c_out, int vd, const int* Cs, int n_Cs) {
    auto conv = [](size_t ci, size_t co, size_t k) { return co * ci * k * k + co; };
    const int num_res = n_Cs - 1, Ch = Cs[num_res - 1];
    size_t s = conv(c_in, Cs[0], 3) + 4 * (size_t)Cs[0] + conv(Ch, c_out, 1) + conv(Ch, 1, 1) + 4 +
               (size_t)64 * N * N + 64 + 64 * (size_t)vd + vd;
    for (int i = 0; i < num_res; ++i) s += conv(Cs[i], Cs[i + 1], 3) + conv(Cs[i + 1], Cs[i + 1], 3) + 8 * (size_t)Cs[i + 1];
    return s;
}

// Two kernels that wait for each other, one on each of two streams: does the GPU run the two side by side?  Kernel
// `me` raises words[me], waits (at most `ticks` of the 100 MHz clock) for words[1 - me] and notes in words[2 + me]
// whether it came (1) or not (2).  The persistent tower / tree pair needs exactly this of its two streams, and what
// two streams get depends on the rest of the process: the runtime hands out four hardware queues per priority and
// shares them out among all streams of that priority (two kernels in one in-order queue never overlap), and the command
// processor serves a normal-priority queue that shares a pipe with a busy high-priority one only when that one idles.
__global__ void az_queue_probe_kernel(unsigned* words, int me, unsigned ticks) {
    if (threadIdx.x != 0) return;
    __hip_atomic_store(words + me, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned r = 2u;
    for (;;) {
        if (__hip_atomic_load(words + 1 - me, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { r = 1u; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)ticks) break;
        __builtin_amdgcn_s_sleep(20);
    }
    __hip_atomic_store(words + 2 + me, r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Both launch orders: the pair's tree kernel may start before the tower kernel or after it.
static int queues_side_by_side(hipStream_t s1, hipStream_t s2, unsigned* words, bool* ok) {
    *ok = true;
    for (int order = 0; order < 2 && *ok; ++order) {
        HIPCHK(hipMemsetAsync(words, 0, 16, s1));
        HIPCHK(hipStreamSynchronize(s1));
        hipLaunchKernelGGL(az_queue_probe_kernel, dim3(1), dim3(64), 0, order ? s2 : s1, words, order ? 1 : 0, 2000000u);   // 20 ms
        hipLaunchKernelGGL(az_queue_probe_kernel, dim3(1), dim3(64), 0, order ? s1 : s2, words, order ? 0 : 1, 2000000u);
        HIPCHK(hipStreamSynchronize(s1));
        HIPCHK(hipStreamSynchronize(s2));
        unsigned h[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpy(h, words, 16, hipMemcpyDeviceToHost));
        *ok = h[2] == 1u && h[3] == 1u;
    }
    return AZ_OK;
}

static int ctx_build(az_ctx* c, const az_config* cfg);
static int pair_warm_up(az_ctx* c);

extern "C" int az_ctx_create(az_ctx** out, const az_config* cfg) {
    if (!out || !cfg) return fail(AZ_E_ARG, "null argument");
    *out = nullptr;
    if (cfg->N < 3 || cfg->N * cfg->N > 512 - 64) return fail(AZ_E_ARG, "board side %d unsupported (3..21)", cfg->N);
    if (cfg->K < 2 || cfg->K > 8 || cfg->K > cfg->N) return fail(AZ_E_ARG, "win length %d unsupported (2..8, <= N)", cfg->K);
    if (cfg->G < 1) return fail(AZ_E_ARG, "G must be >= 1");
    if (cfg->leaves_per_round < 0 || cfg->leaves_per_round > 64)
        return fail(AZ_E_ARG, "leaves_per_round %d out of range (0 = 1 .. 64)", cfg->leaves_per_round);
    if (cfg->eval_mode == AZ_EVAL_TOWER) {
        if (cfg->n_Cs < 2 || cfg->n_Cs > AZ_MAX_LAYERS) return fail(AZ_E_ARG, "n_Cs %d out of range", cfg->n_Cs);
        for (int i = 1; i < cfg->n_Cs; ++i)
            if (cfg->Cs[i] != cfg->Cs[0])
                return fail(AZ_E_ARG, "ResBlock widths must be uniform (x + input, network.hpp:36)");
        if (cfg->precision != AZ_PREC_BF16 && cfg->precision != AZ_PREC_FP16 && cfg->precision != AZ_PREC_FP16X3)
            return fail(AZ_E_ARG, "precision %d unknown (AZ_PREC_FP16, AZ_PREC_BF16, AZ_PREC_FP16X3)", cfg->precision);
        if (cfg->c_in < 0 || cfg->c_in > 16 || cfg->c_out < 0 || cfg->c_out > 128 || cfg->value_dim < 0 || cfg->value_dim > 8 || cfg->value_dim == 1)
            return fail(AZ_E_ARG, "tower shape out of range: c_in %d (<= 16), c_out %d (<= 128), value_dim %d (2..8)", cfg->c_in, cfg->c_out, cfg->value_dim);
    }
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(AZ_E_HIP, "no HIP device: libazhip has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(AZ_E_ARG, "device %d out of range (%d devices)", cfg->device, ndev);
    HIPCHK(hipSetDevice(cfg->device));
    if (cfg->device < AZ_MAX_DEVICES) {                      // once per device, with that device current
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (!g_scratch_ptr[cfg->device]) {
            void* p = nullptr;
            HIPCHK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            g_scratch_ptr[cfg->device] = (unsigned char*)p;
        }
    }
    az_ctx* c = new az_ctx;
    c->cfg = *cfg;
    const int rc = ctx_build(c, cfg);
    if (rc != AZ_OK) {                       // free whatever was allocated before the failure
        const std::string keep = g_err;
        az_ctx_destroy(c);
        g_err = keep;
        return rc;
    }
    *out = c;
    return AZ_OK;
}

static int ctx_build(az_ctx* c, const az_config* cfg) {
    const int N = cfg->N, A = N * N, G = cfg->G;
    c->A = A; c->W = N + 1; c->P = c->W * c->W; c->Gd = N + 2;
    hipStream_t& pooled2 = c->pooled2;                      // (in the context: a build that fails early still hands it back)
    if (cfg->device >= 0 && cfg->device < AZ_MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        auto& pool = g_stream_pool[cfg->device];
        // (a context that can run pairs takes two streams that ran side by side before, if there are such)
        const bool two = cfg->eval_mode == AZ_EVAL_TOWER && (cfg->c_in == 0 || cfg->c_in == 3) && (cfg->c_out == 0 || cfg->c_out == 1) &&
                         (cfg->value_dim == 0 || cfg->value_dim == 2);
        for (size_t i = pool.size(); i-- > 0 && !c->stream;)
            if ((pool[i].second != nullptr) == two) {
                c->stream = pool[i].first; pooled2 = pool[i].second;
                pool.erase(pool.begin() + (long)i);
            }
    }
    if (!c->stream) HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    AzTreeDev& d = c->d;
    memset(&d, 0, sizeof d);
    d.G = G; d.N = N; d.A = A; d.K = cfg->K;
    d.L = cfg->leaves_per_round > 0 ? cfg->leaves_per_round : 1;
    const size_t S = (size_t)G * d.L;                               // evaluation slots per round
    const int mb = cfg->max_budget > 0 ? cfg->max_budget : 800;
    d.node_cap = cfg->node_cap > 0 ? cfg->node_cap : 4 * mb + 64;
    d.edge_cap = (int)std::min<long long>((long long)d.node_cap * A, 2000000000ll);
    d.path_cap = A + 1;
    d.vl = cfg->vl; d.c_puct_i = (int)cfg->c_puct;
    d.noise_mode = cfg->noise_mode; d.seed = cfg->seed; d.slot_base = cfg->game_slot_base;
    d.budget = mb;
    // A tower whose planes / policy channels / value entries are not the m,n,k games' (3, 1, 2) serves
    // az_tower_forward* only: none of the per-game tree, self-play and replay state below is allocated (at the
    // Blokus evaluator's size, N = 20 and G = 2 048, that is about 4 GB of arenas and row buffers).
    c->tower_only = cfg->eval_mode == AZ_EVAL_TOWER &&
                    !((cfg->c_in == 0 || cfg->c_in == 3) && (cfg->c_out == 0 || cfg->c_out == 1) && (cfg->value_dim == 0 || cfg->value_dim == 2));
    AZCHK(dalloc(c, &c->d_stats, 32)); AZCHK(dalloc(c, &c->d_illegal, 4));
    AZCHK(dalloc(c, &c->d_range, 4)); AZCHK(dalloc(c, &c->d_sat, 2));
  if (!c->tower_only) {
#define DA(field, n) AZCHK(dalloc(c, &d.field, (size_t)(n)))
    DA(cells, (size_t)G * A); DA(to_move, G); DA(hist_len, G); DA(hist, (size_t)G * A);
    DA(game_over, G); DA(final_reward, 2 * (size_t)G);
    DA(arena, G);
    for (int a = 0; a < 2; ++a) {
        AZCHK(dalloc(c, &d.edges[a], (size_t)G * d.edge_cap, false));
        AZCHK(dalloc(c, &d.eact[a], (size_t)G * d.edge_cap, false));
    }
    DA(n_nodes, G); DA(n_edges, G); DA(root_n, G); DA(root_hist_len, G); DA(root_hist, (size_t)G * A);
    AZCHK(dalloc(c, &d.queue, (size_t)G * d.node_cap, false));
    DA(status, G); DA(sims_done, G); DA(sims_target, G); DA(pending, S); DA(path_len, S);
    DA(path, S * d.path_cap); DA(pend_result, 2 * S); DA(err, G);
    DA(leaf_planes, S * 3 * A); DA(policy, S * A); DA(value, 2 * S);
    DA(noise, (size_t)G * A); DA(out_policy, (size_t)G * A);
    DA(move_idx, G); DA(avg_len, G); DA(game_count, G); DA(sample_ctr, G); DA(noise_ctr, G);
    DA(row_cells, (size_t)G * A * A); DA(row_tomove, (size_t)G * A); DA(row_policy, (size_t)G * A * A);
    d.ring_cap = std::max(4 * G * A / 2, 4096);
    DA(counters, 16);
    for (int r = 0; r < 2; ++r) {
        AZCHK(dalloc(c, &c->rings[r].cells, (size_t)d.ring_cap * A)); AZCHK(dalloc(c, &c->rings[r].tomove, (size_t)d.ring_cap));
        AZCHK(dalloc(c, &c->rings[r].policy, (size_t)d.ring_cap * A)); AZCHK(dalloc(c, &c->rings[r].reward, 2 * (size_t)d.ring_cap));
    }
    d.ring_cells = c->rings[0].cells; d.ring_tomove = c->rings[0].tomove; d.ring_policy = c->rings[0].policy; d.ring_reward = c->rings[0].reward;
    d.ring_ctr = d.counters + 4;
    HIPCHK(hipEventCreateWithFlags(&c->flip_ev, hipEventDisableTiming)); DA(g_ctr, (size_t)G * AZ_GC_STRIDE); DA(g_hw_nodes, G); DA(g_hw_edges, G);
#undef DA
    HIPCHK(hipMemset(d.root_hist_len, 0xFF, sizeof(int) * G));      // -1: no tree
    {   // uniform noise default so an unset buffer is still a distribution
        std::vector<float> nz((size_t)G * A, 1.0f / (float)A);
        HIPCHK(hipMemcpy(d.noise, nz.data(), nz.size() * 4, hipMemcpyHostToDevice));
    }
    AZCHK(dalloc(c, &c->pair.plane_seq, S)); AZCHK(dalloc(c, &c->pair.eval_seq, S));
    AZCHK(dalloc(c, &c->pair.abort, 8)); AZCHK(dalloc(c, &c->pair.start, 4)); AZCHK(dalloc(c, &c->pair.busy, G));
    c->pair.plane_stride = (3 * A + 15) / 16 * 16;
    AZCHK(dalloc(c, &c->pair.planes_wt, S * (size_t)(c->pair.plane_stride / 4)));
    if (cfg->eval_mode == AZ_EVAL_TOWER) {
        // The pair's tree kernel gets a stream of its own, made here and not at the first pair launch: making a stream
        // makes a hardware queue, for which the driver takes every queue of the process off the GPU and puts it back --
        // running kernels are saved and restored in the middle of another context's launch.  Which stream: one on which
        // a kernel runs side by side with a kernel on the context's stream, found by trying (queues_side_by_side); the
        // runtime shares four hardware queues out among the streams of a priority, and two kernels in one in-order
        // queue never overlap.  The streams that failed are kept until the search is over, so that the runtime hands
        // the next one another queue.  Same priority as the context's stream, on purpose: the command processor serves
        // a normal-priority queue that shares a pipe with a busy high-priority one only when that one idles, and which
        // queues share a pipe changes with every queue the process makes -- with the tree kernel on a high-priority
        // stream, pairs that had passed this test at creation later ran one kernel at a time until the time-out.
        // No such stream: the context plays lockstep rounds only (az_selfplay_run_pair says why).
        HIPCHK(hipEventCreateWithFlags(&c->pair_join, hipEventDisableTiming));
        std::vector<hipStream_t> failed;
        const bool no_probe = getenv("AZ_PAIR_NO_QUEUE_PROBE") != nullptr;          // (queue experiments)
        for (int attempt = 0; attempt < 8 && !c->stream2; ++attempt) {
            hipStream_t s2 = pooled2;
            pooled2 = nullptr;
            if (!s2 && hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) {
                for (hipStream_t f : failed) (void)hipStreamDestroy(f);
                return fail(AZ_E_HIP, "no stream for the pair's tree kernel: %s", hipGetErrorString(hipGetLastError()));
            }
            bool ok = true;
            const int rc = no_probe ? (int)AZ_OK : queues_side_by_side(c->stream, s2, c->pair.start, &ok);
            if (rc != AZ_OK) { (void)hipStreamDestroy(s2); for (hipStream_t f : failed) (void)hipStreamDestroy(f); return rc; }
            if (ok) { c->stream2 = s2; c->pair_queue_attempts = attempt + 1; }
            else failed.push_back(s2);
        }
        for (hipStream_t f : failed) (void)hipStreamDestroy(f);
        HIPCHK(hipMemset(c->pair.start, 0, 16));
    }
    {
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
        c->n_cu = prop.multiProcessorCount;
    }
    AZCHK(dalloc(c, &c->d_actions, G)); AZCHK(dalloc(c, &c->d_reward, 2 * (size_t)G));
    AZCHK(dalloc(c, &c->d_done, G));
    AZCHK(dalloc(c, &c->d_export, (size_t)G * 3 * A));
    if (az_tree_configure(&d) != 0) return fail(AZ_E_HIP, "tree kernel LDS configuration failed (node_cap %d)", d.node_cap);
  }

    c->Bmax = (int)S;
    if (cfg->eval_mode == AZ_EVAL_TOWER) {
        c->C = cfg->Cs[0];
        c->C_pad = (c->C + 127) / 128 * 128;
        c->num_res = cfg->n_Cs - 1;
        c->prec = cfg->precision == AZ_PREC_BF16 ? 0 : 1;           // kernel template index: 0 bf16, 1 fp16
        c->split = cfg->precision == AZ_PREC_FP16X3;
        c->winfo.precision_in_use = cfg->precision;
        c->c_in = cfg->c_in > 0 ? cfg->c_in : 3; c->c_out = cfg->c_out > 0 ? cfg->c_out : 1; c->vd = cfg->value_dim > 0 ? cfg->value_dim : 2;
        const bool mnk_shape = c->c_in == 3 && c->c_out == 1 && c->vd == 2;
        // whole tower in one launch when a board fits one 256-row LDS tile and the bias table of all
        // layers still fits beside it (160 KB per CU); otherwise one launch per convolution
        c->fused = mnk_shape && !c->split && c->C_pad == 128 && c->P <= 256 && !getenv("AZ_NO_FUSED") &&
                   az_tower_fused_lds_bytes(N, 2 * c->num_res, 256) <= 160 * 1024;
        c->conv_kind = az_conv_kind();
        c->conv_layer_elems = (size_t)(c->C_pad / 128) * (c->C_pad / 128) * 18 * 128 * 64;
        AZCHK(dalloc(c, &c->stem_b, c->C_pad));
        AZCHK(dalloc(c, &c->conv_b, (size_t)2 * c->num_res * c->C_pad));
        if (c->fused) {                 // whole tower in one launch: activations never leave LDS
            AZCHK(dalloc(c, &c->stem_wbf, (size_t)128 * 32));
            AZCHK(dalloc(c, &c->conv_wf, (size_t)2 * c->num_res * 9 * 128 * 128));
            AZCHK(dalloc(c, &c->head_pt, (size_t)258 + 64 * (size_t)A + 64 + 128 + 2));
        } else {                        // per-layer path: two activation images in HBM
            const size_t M_pad = ((size_t)c->Bmax * c->P + 255) / 256 * 256;
            c->act_rows = M_pad + 2 * (size_t)c->Gd;
            AZCHK(dalloc(c, &c->actX, c->act_rows * c->C_pad));
            AZCHK(dalloc(c, &c->actH, c->act_rows * c->C_pad));
            AZCHK(dalloc(c, &c->actR, c->act_rows * c->C_pad));
            // The residual stream between the blocks as a pair of 16-bit images (the 16-bit copy the next convolution
            // reads anyway + the remainder): 8 instead of 10 bytes per element through HBM in a block's second
            // convolution, which is HBM-bound at 128 channels.  The default per-layer kernel only; AZ_NO_HILO keeps
            // the fp32 image between the blocks too (A/B).
            // At 256 channels that convolution is MFMA-bound and the pair buys nothing (AZ_HILO=1 forces it for the A/B).
            c->hilo = c->conv_kind == 0 && !c->split && !getenv("AZ_NO_HILO") && (c->C_pad == 128 || getenv("AZ_HILO"));
            if (c->hilo) AZCHK(dalloc(c, &c->actL, c->act_rows * c->C_pad));
            if (az_tower_heads_lds_bytes(N, c->C_pad, c->c_out) > 128 * 1024 || az_tower_stem_lds_bytes(N, c->C_pad, c->c_in) > 64 * 1024)
                return fail(AZ_E_ARG, "tower heads do not fit LDS: c_out %d x %d channels, c_in %d", c->c_out, c->C_pad, c->c_in);
            AZCHK(dalloc(c, &c->stem_w, (size_t)c->C_pad * c->c_in * 9));
            AZCHK(dalloc(c, &c->conv_w, c->conv_layer_elems * 2 * c->num_res));
            if (c->split) {
                if (c->conv_kind != 0) return fail(AZ_E_ARG, "AZ_PREC_FP16X3 runs on the default per-layer kernel only (unset AZ_T256_32 / AZ_CONV_T128)");
                AZCHK(dalloc(c, &c->catX, c->act_rows * c->C_pad * 3));
                AZCHK(dalloc(c, &c->catH, c->act_rows * c->C_pad * 3));
                if (getenv("AZ_X3_SEPARATE_SPLIT")) AZCHK(dalloc(c, &c->actS, c->act_rows * c->C_pad));
                AZCHK(dalloc(c, &c->conv_w3, c->conv_layer_elems * 3 * 2 * c->num_res));
            }
            AZCHK(dalloc(c, &c->head_p, (size_t)(c->c_out + 1) * c->C_pad + c->c_out + 1 + 64 * (size_t)A + 64 + 64 * (size_t)c->vd + c->vd));
        }
        AZCHK(dalloc(c, &c->blob_stage, weights_size(N, c->c_in, c->c_out, c->vd, cfg->Cs, cfg->n_Cs), false));
        AZCHK(dalloc(c, &c->tower_in, (size_t)c->Bmax * c->c_in * A));
        AZCHK(dalloc(c, &c->tower_pol, (size_t)c->Bmax * c->c_out * A)); AZCHK(dalloc(c, &c->tower_val, (size_t)c->vd * c->Bmax));
        if (az_tower_configure() != 0) return fail(AZ_E_HIP, "tower kernel LDS configuration failed");
    }
    HIPCHK(hipDeviceSynchronize());
    AZCHK(pair_warm_up(c));
    return AZ_OK;
}

extern "C" int az_ctx_destroy(az_ctx* c) {
    if (!c) return AZ_OK;
    (void)hipSetDevice(c->cfg.device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    if (c->cfg.device >= 0 && c->cfg.device < AZ_MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        if (g_scratch_user[c->cfg.device] == c->stream) g_scratch_user[c->cfg.device] = nullptr;
    }
    for (void* p : c->allocs) (void)hipFree(p);
    for (auto& e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->flip_ev) (void)hipEventDestroy(c->flip_ev);
    if (c->pair_join) (void)hipEventDestroy(c->pair_join);
    if (c->cfg.device >= 0 && c->cfg.device < AZ_MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_pair_mu[c->cfg.device]);
        if (g_pair_last[c->cfg.device] == c) g_pair_last[c->cfg.device] = nullptr;
    }
    if (c->stream && c->cfg.device >= 0 && c->cfg.device < AZ_MAX_DEVICES) {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        g_stream_pool[c->cfg.device].push_back({c->stream, c->stream2 ? c->stream2 : c->pooled2});
    } else {
        if (c->stream2) (void)hipStreamDestroy(c->stream2);
        if (c->stream) (void)hipStreamDestroy(c->stream);
    }
    delete c;
    return AZ_OK;
}

extern "C" int az_device_count(void) {
    int n = 0;
    HIPCHK(hipGetDeviceCount(&n));
    return n;
}
extern "C" int az_tower_is_fused(const az_ctx* c) { return c && c->fused ? 1 : 0; }
extern "C" int az_ctx_cells(const az_ctx* c) { return c ? c->A : 0; }
extern "C" int az_ctx_device(const az_ctx* c) { return c ? c->cfg.device : -1; }
extern "C" int az_tower_shape(const az_ctx* c, int* N, int* c_in, int* c_out, int* value_dim, int* max_batch) {
    NEED(c);
    if (c->cfg.eval_mode != AZ_EVAL_TOWER) return fail(AZ_E_STATE, "context has no tower");
    if (N) *N = c->cfg.N;
    if (c_in) *c_in = c->c_in;
    if (c_out) *c_out = c->c_out;
    if (value_dim) *value_dim = c->vd;
    if (max_batch) *max_batch = c->Bmax;
    return AZ_OK;
}
extern "C" int az_selfplay_batch(const az_ctx* c) { return c ? c->cfg.G * c->d.L : 0; }
static int pair_check(az_ctx* c);
extern "C" int az_sync(az_ctx* c) { NEED(c); AZCHK(sync_all(c)); return c->tower_only ? AZ_OK : pair_check(c); }
extern "C" void* az_stream(az_ctx* c) { return c ? (void*)c->stream : nullptr; }

// ----------------------------------------------------------------- weights
extern "C" size_t az_weights_size(const az_ctx* c) {
    if (!c || c->cfg.eval_mode != AZ_EVAL_TOWER) return 0;
    return weights_size(c->cfg.N, c->c_in, c->c_out, c->vd, c->cfg.Cs, c->cfg.n_Cs);
}

// load_network (util.h:60-105) minus the file: the blob is staged on the device, validated there, and
// folded / packed by the kernels of weights.hip on the context's stream -- in stream order with the
// evaluator launches, so weights can be swapped between two rounds of a running self-play.
static int load_weights_staged(az_ctx* c) {
    const int N = c->cfg.N, A = N * N, C = c->C;
    AzFoldArgs a;
    memset(&a, 0, sizeof a);
    a.blob = c->blob_stage; a.n_floats = az_weights_size(c);
    a.N = N; a.A = A; a.C = C; a.Cp = c->C_pad; a.nres = c->num_res;
    // every load starts from the configured operand format (an earlier blob may have put the context on bf16)
    c->prec = c->cfg.precision == AZ_PREC_BF16 ? 0 : 1;
    a.fused = c->fused ? 1 : 0; a.prec = c->prec; a.layer16 = c->conv_kind == 0;
    a.c_in = c->c_in; a.c_out = c->c_out; a.vd = c->vd;
    size_t o = 0;                                            // module registration order: CBlock, PHead, VHead, RBlock_i
    a.off_stem_w = o; o += (size_t)C * c->c_in * 9 + C;
    a.off_stem_bn = o; o += 4 * (size_t)C;
    a.off_pc = o; o += (size_t)C * c->c_out + c->c_out;
    a.off_vc = o; o += (size_t)C + 1;
    a.off_vbn = o; o += 4;
    a.off_fc1 = o; o += (size_t)64 * A + 64 + 64 * (size_t)c->vd + c->vd;
    a.off_res0 = o; a.res_stride = 2 * ((size_t)9 * C * C + C) + 8 * (size_t)C;
    if (o + (size_t)c->num_res * a.res_stride != a.n_floats) return fail(AZ_E_STATE, "blob layout does not add up");
    a.stem_w = c->stem_w; a.stem_b = c->stem_b; a.stem_wbf = c->stem_wbf; a.conv_wf = c->conv_wf; a.conv_w = c->conv_w;
    a.conv_layer_elems = c->conv_layer_elems; a.conv_b = c->conv_b; a.head_p = c->head_p; a.head_pt = c->head_pt;
    a.conv_w3 = c->conv_w3;
    // validate before anything is overwritten: a bad blob leaves the current weights in place
    HIPCHK(hipMemsetAsync(c->d_illegal, 0, 16, c->stream));
    az_launch_check_blob(c->stream, &a, c->d_illegal);
    int bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, c->d_illegal, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (bad & 1) return fail(AZ_E_ARG, "weight blob holds non-finite values");
    if (bad & 2) return fail(AZ_E_ARG, "weight blob has a BatchNorm running_var + 1e-5 <= 0: its channel cannot be folded");
    // where the folded weights sit in fp16's range: a dry run of the fold, still before anything is overwritten
    {
        const int init[4] = {0, 0, 0, 0x7F800000};
        int rng[4];
        HIPCHK(hipMemcpyAsync(c->d_range, init, sizeof init, hipMemcpyHostToDevice, c->stream));
        az_launch_fold_range(c->stream, &a, c->d_range);
        HIPCHK(hipMemcpyAsync(rng, c->d_range, sizeof rng, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        az_weights_info& wi = c->winfo;
        wi.precision_in_use = c->cfg.precision; wi.fell_back = 0;
        wi.overflow_weights = rng[0]; wi.underflow_channels = rng[1];
        memcpy(&wi.max_abs_weight, &rng[2], 4); memcpy(&wi.min_channel_max, &rng[3], 4);
        if (rng[3] == 0x7F800000) wi.min_channel_max = 0.f;
        if (c->prec == 1 && (rng[0] > 0 || rng[1] > 0)) {
            if (!c->cfg.range_fallback || c->split)
                return fail(AZ_E_ARG, "folded weights leave fp16's range: %d beyond +-65504 (largest %.4g), %d output channels whose "
                            "largest weight is below 2^-14 (smallest %.4g); load with AZ_PREC_BF16, or set az_config.range_fallback",
                            rng[0], wi.max_abs_weight, rng[1], wi.min_channel_max);
            c->prec = 0; a.prec = 0;                         // this blob runs on bf16 operands
            wi.precision_in_use = AZ_PREC_BF16; wi.fell_back = 1;
        }
    }
    az_launch_fold(c->stream, &a);                           // padding entries of the images stay at their initial zero
    HIPCHK(hipGetLastError());
    c->have_weights = true;
    return AZ_OK;
}

static int load_weights_common(az_ctx* c, const float* blob, size_t n_floats, hipMemcpyKind kind) {
    NEED(c); NEED(blob);
    if (c->cfg.eval_mode != AZ_EVAL_TOWER) return fail(AZ_E_STATE, "context has no tower");
    if (n_floats != az_weights_size(c)) return fail(AZ_E_ARG, "blob has %zu floats, expected %zu", n_floats, az_weights_size(c));
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipMemcpyAsync(c->blob_stage, blob, n_floats * 4, kind, c->stream));
    return load_weights_staged(c);                           // synchronises the stream: the caller's blob is free on return
}

extern "C" int az_load_weights(az_ctx* c, const float* blob, size_t n_floats) {
    return load_weights_common(c, blob, n_floats, hipMemcpyHostToDevice);
}

// The blob already on this GPU (after an RCCL broadcast): nothing crosses PCIe.  The caller makes sure the
// blob is complete (its producer's stream synchronised) before the call.
extern "C" int az_load_weights_dev(az_ctx* c, const float* blob_dev, size_t n_floats) {
    return load_weights_common(c, blob_dev, n_floats, hipMemcpyDeviceToDevice);
}

extern "C" int az_weights_info_get(const az_ctx* c, az_weights_info* out) {
    NEED(c); NEED(out);
    if (c->cfg.eval_mode != AZ_EVAL_TOWER) return fail(AZ_E_STATE, "context has no tower");
    *out = c->winfo;
    return AZ_OK;
}

// ------------------------------------------------------------------- tower
// actX / actH / actR: the activation images of the per-layer path ([guard + M_pad + guard][C_pad] rows, guard
// and padding rows zero): 16-bit inputs of a block's two convolutions and the fp32 residual stream, which every
// block's second convolution updates in place and the heads read; unused by the fused kernel, whose
// activations never leave LDS and registers.
static int tower_launch(az_ctx* c, const uint8_t* planes_dev, int B, float* pol_dev, float* val_dev,
                        unsigned short* actX, unsigned short* actH, float* actR, unsigned short* actL, bool sampled,
                        const unsigned* mask = nullptr) {
    const int N = c->cfg.N, Cp = c->C_pad, Gd = c->Gd, M = B * c->P;
    c->other_work = true;
    if (c->fused) {
        const bool on = prof_begin(c, 0, sampled, c->stream);
        az_launch_tower_fused(c->stream, c->prec, planes_dev, c->stem_wbf, c->stem_b, c->conv_wf, c->conv_b,
                              c->head_pt, pol_dev, val_dev, B, N, Gd, 2 * c->num_res, c->d_sat);
        prof_end(c, on, c->stream);
        HIPCHK(hipGetLastError());
        return AZ_OK;
    }
    az_launch_stem(c->stream, c->prec, planes_dev, actX, actR, c->hilo ? actL : nullptr, c->stem_w, c->stem_b, B, N, Cp, Gd, c->c_in);
    if (c->split) {
        // AZ_PREC_FP16X3.  Every operand is a pair hi + lo of fp16 values; conv(W, x) = W_hi x_hi + W_hi x_lo + W_lo x_hi
        // as ONE reduction over three times the channels (activation rows [hi | lo | hi], weight image [W_hi | W_hi |
        // W_lo], see az_cat3_kernel) on the per-layer path's own kernel, whose epilogue splits the fp32 layer output
        // into the next convolution's [hi | lo | hi] rows (az_launch_conv3x3_cat); the residual stream stays one fp32
        // image, updated in place.  AZ_X3_SEPARATE_SPLIT=1 keeps the first form for the A/B: fp32 layer outputs through
        // the residual form (on a zeroed image for a block's first convolution), split by az_cat3_kernel -- the same
        // arithmetic, bit for bit, with a memset and two more passes over HBM per block.
        const size_t M_pad = ((size_t)M + 255) / 256 * 256;
        const size_t body = (size_t)Gd * Cp;                                      // the rows between the guards
        const bool separate = c->actS != nullptr;                                   // (AZ_X3_SEPARATE_SPLIT when the context was made)
        az_launch_cat3(c->stream, c->prec, actR + body, c->catX + 3 * body, M_pad, Cp);
        for (int i = 0; i < c->num_res; ++i) {
            const unsigned short* w1 = c->conv_w3 + (size_t)(2 * i) * 3 * c->conv_layer_elems;
            const unsigned short* w2 = w1 + 3 * c->conv_layer_elems;
            const float* b1 = c->conv_b + (size_t)(2 * i) * Cp;
            if (separate) HIPCHK(hipMemsetAsync(c->actS + body, 0, M_pad * Cp * 4, c->stream));
            bool on = prof_begin(c, 0, sampled, c->stream);
            if (separate) az_launch_conv3x3(c->stream, c->prec, 0, c->catX, actH, c->actS, nullptr, w1, b1, Cp, M, N, Gd, 1, 3 * Cp);
            else az_launch_conv3x3_cat(c->stream, c->prec, c->catX, c->catH, nullptr, w1, b1, Cp, M, N, Gd, 3 * Cp);
            prof_end(c, on, c->stream);
            if (separate) az_launch_cat3(c->stream, c->prec, c->actS + body, c->catH + 3 * body, M_pad, Cp);
            on = prof_begin(c, 0, sampled, c->stream);
            // (the last block's output feeds the heads only: they read the fp32 stream)
            if (separate || i + 1 == c->num_res) az_launch_conv3x3(c->stream, c->prec, 0, c->catH, actX, actR, nullptr, w2, b1 + Cp, Cp, M, N, Gd, 1, 3 * Cp);
            else az_launch_conv3x3_cat(c->stream, c->prec, c->catH, c->catX, actR, w2, b1 + Cp, Cp, M, N, Gd, 3 * Cp);
            prof_end(c, on, c->stream);
            if (separate && i + 1 < c->num_res) az_launch_cat3(c->stream, c->prec, actR + body, c->catX + 3 * body, M_pad, Cp);
        }
        az_launch_heads(c->stream, c->prec, actR, c->head_p, pol_dev, val_dev, B, N, Cp, Gd, c->c_out, c->vd, c->d_sat, mask);
        HIPCHK(hipGetLastError());
        return AZ_OK;
    }
    for (int i = 0; i < c->num_res; ++i) {
        const unsigned short* w1 = c->conv_w + (size_t)(2 * i) * c->conv_layer_elems;
        const unsigned short* w2 = c->conv_w + (size_t)(2 * i + 1) * c->conv_layer_elems;
        bool on = prof_begin(c, 0, sampled, c->stream);
        az_launch_conv3x3(c->stream, c->prec, c->conv_kind, actX, actH, nullptr, nullptr, w1, c->conv_b + (size_t)(2 * i) * Cp, Cp, M, N, Gd, 1, 0);
        prof_end(c, on, c->stream);
        on = prof_begin(c, 0, sampled, c->stream);
        // the stream stays a 16-bit pair from block to block; the last block leaves it in fp32 for the heads
        az_launch_conv3x3(c->stream, c->prec, c->conv_kind, actH, actX, c->hilo && i + 1 < c->num_res ? nullptr : actR,
                          c->hilo ? actL : nullptr, w2, c->conv_b + (size_t)(2 * i + 1) * Cp, Cp, M, N, Gd, 1, 0);
        prof_end(c, on, c->stream);
    }
    az_launch_heads(c->stream, c->prec, actR, c->head_p, pol_dev, val_dev, B, N, Cp, Gd, c->c_out, c->vd, c->d_sat, mask);
    HIPCHK(hipGetLastError());
    return AZ_OK;
}

static int tower_run(az_ctx* c, const uint8_t* planes_dev, int B, float* pol_dev, float* val_dev, const unsigned* mask = nullptr) {
    if (!c->have_weights) return fail(AZ_E_NOWEIGHTS, "az_load_weights has not been called");
    if (B < 1 || B > c->Bmax) return fail(AZ_E_ARG, "batch %d out of range (1..%d)", B, c->Bmax);
    const bool sampled = c->prof_every > 0 && (c->prof_calls++ % c->prof_every) == 0;
    return tower_launch(c, planes_dev, B, pol_dev, val_dev, c->actX, c->actH, c->actR, c->actL, sampled, mask);
}

extern "C" int az_tower_forward_dev(az_ctx* c, const uint8_t* boards_dev, int B, float* policy_dev, float* value_dev) {
    NEED(c); NEED(boards_dev); NEED(policy_dev); NEED(value_dev);
    if (c->cfg.eval_mode != AZ_EVAL_TOWER) return fail(AZ_E_STATE, "context has no tower");
    HIPCHK(hipSetDevice(c->cfg.device));
    return tower_run(c, boards_dev, B, policy_dev, value_dev);
}

// The evaluator as the search needs it: only the policy entries a leaf's expansion will read.  The fused kernel
// (c_out = 1) writes its N * N entries in any case.
extern "C" int az_tower_forward_masked_dev(az_ctx* c, const uint8_t* boards_dev, int B, float* policy_dev, float* value_dev,
                                           const uint32_t* wanted_dev) {
    NEED(c); NEED(boards_dev); NEED(policy_dev); NEED(value_dev); NEED(wanted_dev);
    if (c->cfg.eval_mode != AZ_EVAL_TOWER) return fail(AZ_E_STATE, "context has no tower");
    HIPCHK(hipSetDevice(c->cfg.device));
    return tower_run(c, boards_dev, B, policy_dev, value_dev, c->fused ? nullptr : (const unsigned*)wanted_dev);
}

extern "C" int az_tower_forward(az_ctx* c, const uint8_t* boards, int B, float* policy, float* value) {
    NEED(c); NEED(boards); NEED(policy); NEED(value);
    if (c->cfg.eval_mode != AZ_EVAL_TOWER) return fail(AZ_E_STATE, "context has no tower");
    if (B < 1 || B > c->Bmax) return fail(AZ_E_ARG, "batch %d out of range (1..%d)", B, c->Bmax);
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t A = c->A;
    HIPCHK(hipMemcpyAsync(c->tower_in, boards, (size_t)B * c->c_in * A, hipMemcpyHostToDevice, c->stream));
    AZCHK(tower_run(c, c->tower_in, B, c->tower_pol, c->tower_val));
    HIPCHK(hipMemcpyAsync(policy, c->tower_pol, (size_t)B * c->c_out * A * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(value, c->tower_val, (size_t)B * c->vd * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AZ_OK;
}

// az_tower_forward_dev for callers that cannot allocate: on a per-layer context the two activation images
// live in this GPU's `scratch` array instead of the context's own buffers, so the batch is bounded by the
// array, not by az_config.G.  No hipMalloc / hipFree, no host allocation, no event.  Every call on a GPU
// shares that GPU's array: calls that overlap on one GPU race.  The array keeps the last call's contents, so
// the rows that must read as zero (the guard rows at both ends and the rows behind the last board) are
// cleared on every call.
extern "C" size_t az_tower_static_max_batch(const az_ctx* c) {
    if (!c || c->cfg.eval_mode != AZ_EVAL_TOWER) return 0;
    if (c->fused) return (size_t)c->Bmax;                    // the fused kernel has no scratch: the context's own limit
    // the array holds three images of the same rows: two 16-bit ones (a quarter each) and the fp32 one (a half)
    const size_t row = (size_t)c->C_pad * 2, fixed = 2 * (size_t)c->Gd + 255;   // guard rows + round-up to a 256-row tile
    const size_t rows = AZ_STATIC_SCRATCH_BYTES / (c->hilo ? 5 : 4) / row;      // two 16-bit images + fp32 (+ the 16-bit remainder)
    return rows > fixed ? (rows - fixed) / (size_t)c->P : 0;
}

extern "C" int az_tower_forward_static(az_ctx* c, const uint8_t* boards_dev, int B, float* policy_dev, float* value_dev) {
    NEED(c); NEED(boards_dev); NEED(policy_dev); NEED(value_dev);
    if (c->cfg.eval_mode != AZ_EVAL_TOWER) return fail(AZ_E_STATE, "context has no tower");
    if (!c->have_weights) return fail(AZ_E_NOWEIGHTS, "az_load_weights has not been called");
    if (c->split) return fail(AZ_E_STATE, "az_tower_forward_static does not run AZ_PREC_FP16X3 (its activation images are three times as wide; use az_tower_forward_dev)");
    HIPCHK(hipSetDevice(c->cfg.device));
    if (c->fused) {                                          // activations stay in LDS: nothing to place
        if (B < 1 || B > c->Bmax) return fail(AZ_E_ARG, "batch %d out of range (1..%d)", B, c->Bmax);
        return tower_launch(c, boards_dev, B, policy_dev, value_dev, nullptr, nullptr, nullptr, nullptr, false);
    }
    unsigned char* const base = c->cfg.device < AZ_MAX_DEVICES ? g_scratch_ptr[c->cfg.device] : nullptr;
    if (!base) return fail(AZ_E_STATE, "static scratch of device %d was not set up", c->cfg.device);
    const size_t maxB = az_tower_static_max_batch(c);
    if (B < 1 || (size_t)B > maxB)
        return fail(AZ_E_CAPACITY, "batch %d does not fit the static scratch: at most %zu boards of %dx%d with %d channels (%u MiB per GPU)",
                    B, maxB, c->cfg.N, c->cfg.N, c->C_pad, AZ_STATIC_SCRATCH_BYTES >> 20);
    {   // One array per GPU, whoever calls: work of another context (another stream) that still uses it would be
        // overwritten.  Calls on one stream are ordered by the stream; a different stream must have drained first.
        std::lock_guard<std::mutex> lk(g_scratch_mu);
        hipStream_t& user = g_scratch_user[c->cfg.device];
        if (user && user != c->stream && hipStreamQuery(user) == hipErrorNotReady)
            return fail(AZ_E_STATE, "the static scratch of GPU %d is in use by another context's az_tower_forward_static call that has "
                                    "not finished (one array per GPU: synchronise that context first)", c->cfg.device);
        user = c->stream;
    }
    const size_t row = (size_t)c->C_pad * 2;
    const size_t M = (size_t)B * c->P, M_pad = (M + 255) / 256 * 256, rows = M_pad + 2 * (size_t)c->Gd;
    // the array in equal parts: 16-bit image, 16-bit image, fp32 image (two parts), and the 16-bit remainder of the
    // stream where the context keeps it as a pair
    const size_t part = (AZ_STATIC_SCRATCH_BYTES / (c->hilo ? 5 : 4)) & ~(size_t)255;
    unsigned char* const img[4] = {base, base + part, base + 2 * part, base + 4 * part};
    for (int k = 0; k < (c->hilo ? 4 : 3); ++k) {
        const size_t rb = k == 2 ? 2 * row : row;            // bytes per row: 16-bit images, fp32 residual stream
        HIPCHK(hipMemsetAsync(img[k], 0, (size_t)c->Gd * rb, c->stream));
        HIPCHK(hipMemsetAsync(img[k] + ((size_t)c->Gd + M) * rb, 0, (rows - c->Gd - M) * rb, c->stream));
    }
    return tower_launch(c, boards_dev, B, policy_dev, value_dev, (unsigned short*)img[0], (unsigned short*)img[1], (float*)img[2],
                        c->hilo ? (unsigned short*)img[3] : nullptr, false);
}

// --------------------------------------------------------------------- env
extern "C" int az_env_reset(az_ctx* c) {
    NEED(c); NEED_GAMES(c);
    HIPCHK(hipSetDevice(c->cfg.device));
    az_launch_env_reset(c->stream, &c->d);
    HIPCHK(hipStreamSynchronize(c->stream));
    return AZ_OK;
}

extern "C" int az_env_step(az_ctx* c, const int32_t* actions, float* reward, uint8_t* done) {
    NEED(c); NEED_GAMES(c); NEED(actions); NEED(reward); NEED(done);
    HIPCHK(hipSetDevice(c->cfg.device));
    const int G = c->cfg.G;
    HIPCHK(hipMemcpyAsync(c->d_actions, actions, sizeof(int) * G, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->d_illegal, 0, 16, c->stream));
    az_launch_env_step(c->stream, &c->d, c->d_actions, c->d_reward, c->d_done, c->d_illegal);
    int illegal = 0;
    HIPCHK(hipMemcpyAsync(reward, c->d_reward, sizeof(float) * 2 * G, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(done, c->d_done, G, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&illegal, c->d_illegal, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (illegal) return fail(AZ_E_ARG, "%d illegal action(s): occupied cell, out of range, or finished game", illegal);
    return AZ_OK;
}

extern "C" int az_env_boards(az_ctx* c, uint8_t* planes) {
    NEED(c); NEED_GAMES(c); NEED(planes);
    HIPCHK(hipSetDevice(c->cfg.device));
    az_launch_env_export(c->stream, &c->d, c->d_export, nullptr);       // own scratch: never the evaluator's live input
    HIPCHK(hipMemcpyAsync(planes, c->d_export, (size_t)c->cfg.G * 3 * c->A, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AZ_OK;
}

extern "C" int az_env_legal_mask(az_ctx* c, uint8_t* mask) {
    NEED(c); NEED_GAMES(c); NEED(mask);
    HIPCHK(hipSetDevice(c->cfg.device));
    az_launch_env_export(c->stream, &c->d, nullptr, c->d_export);
    HIPCHK(hipMemcpyAsync(mask, c->d_export, (size_t)c->cfg.G * c->A, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AZ_OK;
}

extern "C" int az_env_set(az_ctx* c, const uint8_t* cells, const uint8_t* to_move) {
    NEED(c); NEED_GAMES(c); NEED(cells); NEED(to_move);
    HIPCHK(hipSetDevice(c->cfg.device));
    const int G = c->cfg.G, A = c->A;
    for (size_t i = 0; i < (size_t)G * A; ++i)
        if (cells[i] > 2) return fail(AZ_E_ARG, "game %zu: cell value %d (0 empty, 1, 2)", i / A, cells[i]);
    for (int g = 0; g < G; ++g)
        if (to_move[g] > 1) return fail(AZ_E_ARG, "game %d: side to move %d (0 or 1)", g, to_move[g]);
    HIPCHK(hipStreamSynchronize(c->stream));      // blocking copies below do not order against the stream
    // a position set directly has no move history: it is its own history origin
    std::vector<int> zeros(G, 0);
    std::vector<uint8_t> z8(G, 0);
    HIPCHK(hipMemcpy(c->d.cells, cells, (size_t)G * A, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d.to_move, to_move, G, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d.game_over, z8.data(), G, hipMemcpyHostToDevice));
    // history length = number of stones so that "board full" detection stays right
    std::vector<int> hl(G);
    for (int g = 0; g < G; ++g) { int s = 0; for (int i = 0; i < A; ++i) s += cells[(size_t)g * A + i] != 0; hl[g] = s; }
    HIPCHK(hipMemcpy(c->d.hist_len, hl.data(), sizeof(int) * G, hipMemcpyHostToDevice));
    std::vector<uint16_t> hist((size_t)G * A, 0xFFFF);
    HIPCHK(hipMemcpy(c->d.hist, hist.data(), hist.size() * 2, hipMemcpyHostToDevice));
    return az_tree_clear(c);
}

// Env state of every game from its move history (Env::get_id, mcts.hpp:388):
// replayed on the host with the same rules, then uploaded with the history so
// that find() (node.hpp:83-102) can reuse the tree.
extern "C" int az_env_load(az_ctx* c, const int32_t* hist, const int32_t* hist_len) {
    NEED(c); NEED_GAMES(c); NEED(hist); NEED(hist_len);
    HIPCHK(hipSetDevice(c->cfg.device));
    const int G = c->cfg.G, A = c->A, N = c->cfg.N;
    std::vector<uint8_t> cells((size_t)G * A, 0), tm(G, 0), over(G, 0);
    std::vector<uint16_t> h16((size_t)G * A, 0);
    std::vector<int> hl(G, 0);
    std::vector<float> fr(2 * (size_t)G, 0.f);
    for (int g = 0; g < G; ++g) {
        int to_move = 0;
        if (hist_len[g] < 0 || hist_len[g] > A) return fail(AZ_E_ARG, "game %d: history length %d", g, hist_len[g]);
        for (int t = 0; t < hist_len[g]; ++t) {
            float r[2];
            if (over[g]) return fail(AZ_E_ARG, "game %d: move after the game ended", g);
            const int d = az_rules_step(N, c->cfg.K, cells.data() + (size_t)g * A, &to_move, hist[(size_t)g * A + t], r);
            if (d < 0) return fail(AZ_E_ARG, "game %d: illegal move %d at ply %d", g, hist[(size_t)g * A + t], t);
            over[g] = d == 1; fr[2 * g] = r[0]; fr[2 * g + 1] = r[1];
            h16[(size_t)g * A + t] = (uint16_t)hist[(size_t)g * A + t];
        }
        tm[g] = (uint8_t)to_move; hl[g] = hist_len[g];
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(c->d.cells, cells.data(), cells.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d.to_move, tm.data(), G, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d.game_over, over.data(), G, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d.hist, h16.data(), h16.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d.hist_len, hl.data(), sizeof(int) * G, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(c->d.final_reward, fr.data(), fr.size() * 4, hipMemcpyHostToDevice));
    return AZ_OK;
}

extern "C" int az_set_noise(az_ctx* c, const float* noise) {
    NEED(c); NEED_GAMES(c); NEED(noise);
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(c->d.noise, noise, (size_t)c->cfg.G * c->A * 4, hipMemcpyHostToDevice));
    return AZ_OK;
}

extern "C" int az_get_noise(az_ctx* c, float* noise) {
    NEED(c); NEED_GAMES(c); NEED(noise);
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipMemcpyAsync(noise, c->d.noise, (size_t)c->cfg.G * c->A * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return AZ_OK;
}

extern "C" int az_tree_clear(az_ctx* c) {
    NEED(c); NEED_GAMES(c);
    HIPCHK(hipSetDevice(c->cfg.device));
    const int G = c->cfg.G;
    {   // keep the arena high-water marks of the trees that are about to be dropped
        HIPCHK(hipStreamSynchronize(c->stream));
        std::vector<int> nn(G), ne(G), hn(G), he(G);
        HIPCHK(hipMemcpy(nn.data(), c->d.n_nodes, sizeof(int) * G, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ne.data(), c->d.n_edges, sizeof(int) * G, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(hn.data(), c->d.g_hw_nodes, sizeof(int) * G, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(he.data(), c->d.g_hw_edges, sizeof(int) * G, hipMemcpyDeviceToHost));
        for (int g = 0; g < G; ++g) { hn[g] = std::max(hn[g], nn[g]); he[g] = std::max(he[g], ne[g]); }
        HIPCHK(hipMemcpy(c->d.g_hw_nodes, hn.data(), sizeof(int) * G, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(c->d.g_hw_edges, he.data(), sizeof(int) * G, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemsetAsync(c->d.root_hist_len, 0xFF, sizeof(int) * G, c->stream));
    HIPCHK(hipMemsetAsync(c->d.n_nodes, 0, sizeof(int) * G, c->stream));
    HIPCHK(hipMemsetAsync(c->d.n_edges, 0, sizeof(int) * G, c->stream));
    HIPCHK(hipMemsetAsync(c->d.root_n, 0, sizeof(int) * G, c->stream));
    HIPCHK(hipMemsetAsync(c->d.status, 0, sizeof(int) * G, c->stream));
    HIPCHK(hipMemsetAsync(c->d.pending, 0, sizeof(int) * G * c->d.L, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->search_active = false;
    return AZ_OK;
}

// ------------------------------------------------------------------ search
// evaluate the leaves of games [g0, g0 + count) (slots g*L .. ) on the context's main stream
static int run_evaluator(az_ctx* c, int g0, int count) {
    const size_t s0 = (size_t)g0 * c->d.L;
    const int B = count * c->d.L;
    if (c->cfg.eval_mode == AZ_EVAL_TOWER && !(c->c_in == 3 && c->c_out == 1 && c->vd == 2))
        return fail(AZ_E_STATE, "this context's tower (c_in %d, c_out %d, value_dim %d) does not fit the m,n,k search", c->c_in, c->c_out, c->vd);
    if (c->cfg.eval_mode == AZ_EVAL_TOWER)
        return tower_run(c, c->d.leaf_planes + s0 * 3 * c->A, B, c->d.policy + s0 * c->A, c->d.value + s0 * 2);
    if (c->cfg.eval_mode == AZ_EVAL_HASH) {
        az_launch_eval_hash(c->stream, c->d.leaf_planes + s0 * 3 * c->A, c->d.policy + s0 * c->A, c->d.value + s0 * 2, B, c->A);
        return AZ_OK;
    }
    return fail(AZ_E_STATE, "external evaluator: drive the search with az_search_select/apply");
}
static int run_evaluator(az_ctx* c) { return run_evaluator(c, 0, c->cfg.G); }

static int check_errors(az_ctx* c) {
    std::vector<int> err(c->cfg.G);
    HIPCHK(hipMemcpy(err.data(), c->d.err, sizeof(int) * c->cfg.G, hipMemcpyDeviceToHost));
    for (int g = 0; g < c->cfg.G; ++g)
        if (err[g]) return fail(AZ_E_STATE, "game %d: no selectable child (mcts.hpp:374-375)", g);
    return AZ_OK;
}

// A search from a fresh root expands at most one node per simulation plus the root; self-play searches a new
// game's root twice (warm-up, generate.cpp:77).  An arena that cannot hold that would restart for ever.
static int check_budget(az_ctx* c, int budget, int searches_per_root) {
    const long long need = (long long)searches_per_root * (budget > 1 ? budget : 1) + 2;
    if (need > c->d.node_cap)
        return fail(AZ_E_ARG, "budget %d needs %lld tree nodes per game, the arena holds %d (az_config.node_cap / max_budget)",
                    budget, need, c->d.node_cap);
    return AZ_OK;
}

extern "C" int az_search_begin(az_ctx* c, int budget) {
    NEED(c); NEED_GAMES(c);
    HIPCHK(hipSetDevice(c->cfg.device));
    AZCHK(check_budget(c, budget, 1));
    c->mode = AZ_MODE_SEARCH;
    az_launch_begin(c->stream, &c->d, budget);
    HIPCHK(hipGetLastError());
    c->search_active = true;
    return AZ_OK;
}

extern "C" int az_search(az_ctx* c, int budget, float* out_policy) {
    NEED(c); NEED_GAMES(c);
    if (c->cfg.eval_mode == AZ_EVAL_EXTERNAL) return fail(AZ_E_STATE, "external evaluator: use az_search_select/apply");
    AZCHK(az_search_begin(c, budget));
    const int b = budget > 1 ? budget : 1;
    // every game needs b simulations, up to L per round, plus one root evaluation for a
    // fresh root; round r applies the evaluations of round r-1 and issues those of round r
    const int rounds = (b + c->d.L - 1) / c->d.L + 2;
    for (int r = 0; r < rounds; ++r) {
        az_launch_round(c->stream, &c->d, 3, AZ_MODE_SEARCH, 0, c->cfg.G);
        if (r < rounds - 1) AZCHK(run_evaluator(c));
        c->rounds++;
    }
    // a search can need more rounds than that: with L > 1 a round issues fewer than L descents when two of
    // them meet at one stub, and a tree whose arena filled up starts over from a fresh root.  Keep going
    // until every game reports its search finished.
    std::vector<int> st(c->cfg.G);
    for (int extra = 0;; ++extra) {
        HIPCHK(hipMemcpyAsync(st.data(), c->d.status, sizeof(int) * c->cfg.G, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        bool running = false;
        for (int x : st) running = running || x == AZ_ST_RUN;
        if (!running) break;
        if (extra > 4 * (b + 2)) return fail(AZ_E_STATE, "search did not finish within its round bound");
        for (int r = 0; r < 4; ++r) {
            AZCHK(run_evaluator(c));
            az_launch_round(c->stream, &c->d, 3, AZ_MODE_SEARCH, 0, c->cfg.G);
            c->rounds++;
        }
    }
    HIPCHK(hipGetLastError());
    return az_search_result(c, out_policy, nullptr);
}

extern "C" int az_search_select(az_ctx* c, uint8_t* planes, int32_t* pending, int* n_running) {
    NEED(c); NEED_GAMES(c); NEED(planes); NEED(pending); NEED(n_running);
    if (!c->search_active) return fail(AZ_E_STATE, "az_search_begin has not been called");
    HIPCHK(hipSetDevice(c->cfg.device));
    const int G = c->cfg.G, L = c->d.L;
    az_launch_round(c->stream, &c->d, 2, AZ_MODE_SEARCH, 0, c->cfg.G);
    std::vector<int> st(G);
    HIPCHK(hipMemcpyAsync(planes, c->d.leaf_planes, (size_t)G * L * 3 * c->A, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(pending, c->d.pending, sizeof(int) * G * L, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(st.data(), c->d.status, sizeof(int) * G, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int run = 0;
    for (int g = 0; g < G; ++g) {
        run += st[g] == AZ_ST_RUN;
        if (st[g] != AZ_ST_RUN) for (int k = 0; k < L; ++k) pending[g * L + k] = 0;
    }
    *n_running = run;
    c->rounds++;
    return check_errors(c);
}

extern "C" int az_search_apply(az_ctx* c, const float* policy, const float* value) {
    NEED(c); NEED_GAMES(c); NEED(policy); NEED(value);
    if (!c->search_active) return fail(AZ_E_STATE, "az_search_begin has not been called");
    HIPCHK(hipSetDevice(c->cfg.device));
    const size_t S = (size_t)c->cfg.G * c->d.L;
    HIPCHK(hipMemcpyAsync(c->d.policy, policy, S * c->A * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d.value, value, S * 2 * 4, hipMemcpyHostToDevice, c->stream));
    az_launch_round(c->stream, &c->d, 1, AZ_MODE_SEARCH, 0, c->cfg.G);
    HIPCHK(hipStreamSynchronize(c->stream));
    return AZ_OK;
}

extern "C" int az_search_result(az_ctx* c, float* out_policy, int32_t* root_n) {
    NEED(c); NEED_GAMES(c);
    HIPCHK(hipSetDevice(c->cfg.device));
    const int G = c->cfg.G;
    HIPCHK(hipStreamSynchronize(c->stream));
    AZCHK(check_errors(c));
    std::vector<int> st(G);
    HIPCHK(hipMemcpy(st.data(), c->d.status, sizeof(int) * G, hipMemcpyDeviceToHost));
    for (int g = 0; g < G; ++g)
        if (st[g] == AZ_ST_RUN) return fail(AZ_E_STATE, "game %d has not finished its search", g);
    if (out_policy) HIPCHK(hipMemcpy(out_policy, c->d.out_policy, (size_t)G * c->A * 4, hipMemcpyDeviceToHost));
    if (root_n) HIPCHK(hipMemcpy(root_n, c->d.root_n, sizeof(int) * G, hipMemcpyDeviceToHost));
    c->search_active = false;
    return AZ_OK;
}

extern "C" int az_tree_dump(az_ctx* c, int game, int node_cap, int edge_cap, int32_t* n_nodes, int32_t* n_edges,
                            int32_t* node_first, int32_t* node_count, float* edge_p, int32_t* edge_n,
                            float* edge_v, int32_t* edge_child, int32_t* edge_action) {
    NEED(c); NEED_GAMES(c); NEED(n_nodes); NEED(n_edges);
    if (game < 0 || game >= c->cfg.G) return fail(AZ_E_ARG, "game %d out of range", game);
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    int nn = 0, ne = 0, ar = 0, rhl = -1;
    HIPCHK(hipMemcpy(&nn, c->d.n_nodes + game, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&ne, c->d.n_edges + game, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&ar, c->d.arena + game, 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(&rhl, c->d.root_hist_len + game, 4, hipMemcpyDeviceToHost));
    if (rhl < 0) { nn = 0; ne = 0; }
    *n_nodes = nn; *n_edges = ne;
    if (node_cap == 0 && edge_cap == 0) return AZ_OK;
    if (node_cap < nn || edge_cap < ne) return fail(AZ_E_ARG, "dump buffers too small (%d nodes, %d edges)", nn, ne);
    NEED(node_first); NEED(node_count); NEED(edge_p); NEED(edge_n); NEED(edge_v); NEED(edge_child); NEED(edge_action);
    std::vector<AzEdge> ed(ne);
    std::vector<uint16_t> ea(ne);
    HIPCHK(hipMemcpy(ed.data(), c->d.edges[ar] + (size_t)game * c->d.edge_cap, sizeof(AzEdge) * ne, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ea.data(), c->d.eact[ar] + (size_t)game * c->d.edge_cap, 2 * (size_t)ne, hipMemcpyDeviceToHost));
    for (int i = 0; i < ne; ++i) {
        edge_p[i] = ed[i].p; edge_n[i] = ed[i].n; edge_v[i] = ed[i].v; edge_child[i] = ed[i].child; edge_action[i] = ea[i];
    }
    // On the device a node IS its block of child edges (an edge links to the first edge of its child's
    // block; an expanded node l plies below the root has A - root_history - l children).  The dump
    // numbers the blocks in BFS order from the root (node 0) and rewrites the links to those numbers.
    int count = 0;
    if (nn > 0) { node_first[0] = 0; node_count[0] = c->A - rhl; count = 1; }
    for (int i = 0; i < count; ++i) {
        const int fe = node_first[i], k = node_count[i];
        if (fe < 0 || fe + k > ne) return fail(AZ_E_STATE, "game %d: arena links are corrupt", game);
        for (int e = fe; e < fe + k; ++e)
            if (ed[e].child >= 0) {
                if (count >= nn) return fail(AZ_E_STATE, "game %d: more linked blocks than expanded nodes", game);
                node_first[count] = ed[e].child; node_count[count] = k - 1;
                edge_child[e] = count++;
            }
    }
    if (count != nn) return fail(AZ_E_STATE, "game %d: %d linked blocks but %d expanded nodes", game, count, nn);
    return AZ_OK;
}

// --------------------------------------------------------------- self-play
extern "C" int az_selfplay_begin(az_ctx* c, int budget) {
    NEED(c); NEED_GAMES(c);
    if (c->cfg.eval_mode == AZ_EVAL_EXTERNAL) return fail(AZ_E_STATE, "self-play needs a device evaluator");
    HIPCHK(hipSetDevice(c->cfg.device));
    AZCHK(check_budget(c, budget, 2));
    c->d.budget = budget > 1 ? budget : 1;
    c->mode = AZ_MODE_SELFPLAY;
    if (const char* e = getenv("AZ_CHAIN")) {               // az_selfplay_chain from the environment (host programs, A/B runs)
        const int k = atoi(e);
        if (c->d.L == 1 && k >= 0 && k <= 64) c->d.chain = k;
    }
    HIPCHK(hipMemsetAsync(c->d.counters, 0, 16 * sizeof(unsigned long long), c->stream));
    HIPCHK(hipMemsetAsync(c->d.g_ctr, 0, sizeof(unsigned long long) * AZ_GC_STRIDE * c->cfg.G, c->stream));
    HIPCHK(hipMemsetAsync(c->d.g_hw_nodes, 0, sizeof(int) * c->cfg.G, c->stream));
    HIPCHK(hipMemsetAsync(c->d.g_hw_edges, 0, sizeof(int) * c->cfg.G, c->stream));
    HIPCHK(hipMemsetAsync(c->d.err, 0, sizeof(int) * c->cfg.G, c->stream));
    HIPCHK(hipMemsetAsync(c->pair.abort, 0, 32, c->stream));
    HIPCHK(hipMemsetAsync(c->pair.busy, 0, sizeof(unsigned long long) * c->cfg.G, c->stream));
    c->pair_launches = 0; c->pair_rounds = 0;
    az_launch_selfplay_begin(c->stream, &c->d);
    // issue the first evaluation so that every later round is evaluate -> tree
    az_launch_round(c->stream, &c->d, 2, AZ_MODE_SELFPLAY, 0, c->cfg.G);
    HIPCHK(hipGetLastError());
    c->search_active = false;
    return AZ_OK;
}

// Whether the context's self-play rounds can run as the persistent pair: the one-launch tower, one leaf per tree
// per round, and room for a tree wave on every CU beside the tower's workgroup.
static const char* pair_unfit(const az_ctx* c) {
    if (c->cfg.eval_mode != AZ_EVAL_TOWER || !c->fused) return "the tower does not run as one launch per evaluation";
    if (c->d.L != 1) return "leaves_per_round > 1";
    if (!c->stream2) return "no second stream of this process runs side by side with the context's (hardware queues shared with or starved by other streams)";
    if (az_tower_fused_lds_bytes(c->cfg.N, 2 * c->num_res, 256) + 64 + (size_t)az_tree_smem_bytes(c->A) > 160 * 1024)
        return "no LDS left on a CU for a tree wave beside the tower's workgroup";
    return nullptr;
}

// Both kernels of the pair spill a few registers, and the first kernel with a private segment on a hardware queue makes
// the runtime set that queue's scratch memory up in the middle of the dispatch.  Left to the first real launch, that
// happened with the tower workgroups already spinning on the CUs, and about every other process start in which the
// context's streams shared their hardware queues with streams of the caller's ran into the time-out there.  So a
// kernel with the private segment of the tower kernel (80 bytes a lane) runs once on the context's stream and one with
// the tree kernel's (128) on the other while the context is made, each with its kernel's grid.  (Not the pair
// kernels themselves with no rounds: a profile's per-kernel averages would count those launches.  Not a larger
// segment either: with 208 bytes on both streams the time-outs came back, one in three runs of the pair tests --
// the more scratch the queues of a process hold, the sooner the runtime takes some back from a queue.)
template <int WORDS>
__global__ void __launch_bounds__(256) az_scratch_warm_kernel(unsigned* sink, int n) {
    volatile unsigned buf[WORDS];
    for (int i = 0; i < WORDS; ++i) buf[i] = threadIdx.x + (unsigned)i;
    unsigned sum = 0u;
    for (int i = 0; i < n; ++i) sum += buf[(threadIdx.x + (unsigned)i) % (unsigned)WORDS];
    if (sum == 0xFFFFFFFFu) sink[0] = sum;
}
static int pair_warm_up(az_ctx* c) {
    if (c->tower_only || pair_unfit(c)) return AZ_OK;
    hipLaunchKernelGGL(az_scratch_warm_kernel<16>, dim3(c->n_cu), dim3(256), 0, c->stream, c->pair.start + 2, 3);
    hipLaunchKernelGGL(az_scratch_warm_kernel<28>, dim3(c->n_cu), dim3(64), 0, c->stream2, c->pair.start + 2, 3);
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipStreamSynchronize(c->stream2));
    HIPCHK(hipGetLastError());
    return AZ_OK;
}

// A wait of the pair that ran out (the two kernels were not resident together): sticky, reported by every later call.
static int pair_check(az_ctx* c) {
    if (!c->pair_launches) return AZ_OK;
    unsigned ab[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    HIPCHK(hipMemcpy(ab, c->pair.abort, 32, hipMemcpyDeviceToHost));
    if (ab[0]) return fail(AZ_E_STATE, "a hand-over wait of the persistent tower/tree pair ran out (the %s, index %u, wanted word %u and saw %u; waits saw the clock step %u times: "
                                        "that often the waves were taken off the GPU and put back): "
                                        "the two kernels were not resident together (another process on this GPU, or a profiler that "
                                        "serialises launches); the games of this context are no longer in step -- run such set-ups with "
                                        "az_selfplay_run", ab[1] == 0xC0000000u ? "gate in front of the tree kernel, which waits for the tower workgroups to begin" : (ab[1] >> 31) ? "tree wave of the game" : "tower workgroup",
                                        ab[1] & 0x3FFFFFFFu, ab[2], ab[3], ab[4]);
    return AZ_OK;
}

extern "C" int az_selfplay_chain(az_ctx* c, int max_chained) {
    NEED(c); NEED_GAMES(c);
    if (max_chained < 0 || max_chained > 64) return fail(AZ_E_ARG, "max_chained %d out of range (0..64)", max_chained);
    if (c->d.L != 1 && max_chained) return fail(AZ_E_STATE, "chained terminal simulations need leaves_per_round 1");
    c->d.chain = max_chained;
    return AZ_OK;
}

extern "C" int az_selfplay_run_pair(az_ctx* c, int rounds) {
    NEED(c); NEED_GAMES(c);
    if (c->mode != AZ_MODE_SELFPLAY) return fail(AZ_E_STATE, "az_selfplay_begin has not been called");
    if (const char* why = pair_unfit(c)) return fail(AZ_E_STATE, "az_selfplay_run_pair: %s", why);
    if (!c->have_weights) return fail(AZ_E_NOWEIGHTS, "az_load_weights has not been called");
    if (rounds < 1) return AZ_OK;
    HIPCHK(hipSetDevice(c->cfg.device));
    AzPairDev pr = c->pair;
    pr.base = c->pair_base;
    pr.rounds = rounds;
    const char* to = getenv("AZ_PAIR_TIMEOUT_MS");
    const double ms = to ? atof(to) : 4000.0;
    pr.timeout_ticks = (unsigned)std::min(4.0e9, std::max(1.0, ms) * 1e5);         // 100 MHz ticks
    // timing experiments (wrong games with bit 0 or 2): 1 = the tower never waits, 2 = no early look at the next tile's
    // words, 4 = no tree kernel (with 1: the tower alone; without: the tower's first wait runs out, the drain path)
    const char* dbg_s = getenv("AZ_PAIR_FLAGS");
    int dbg = dbg_s ? atoi(dbg_s) : 0;
    {   // AZ_PAIR_FAIL_AFTER=n (tests of the callers' recovery): the context's launch number n goes out without its tree kernel
        static const char* fa = getenv("AZ_PAIR_FAIL_AFTER");
        if (fa && c->pair_launches == atoll(fa)) dbg |= 4;
    }
    pr.flags = (unsigned)(dbg & 3);
    // The gate in front of the tree kernel runs as soon as the last pair launch's tree waves have left and gives the
    // tower kernel one time-out to begin: a launch goes behind another pair launch of the context or onto an idle
    // stream, never behind evaluations or lockstep rounds that are still running (the host waits for those; mixing
    // the two forms is rare).
    if (c->other_work) {
        if (hipStreamQuery(c->stream) == hipErrorNotReady) HIPCHK(hipStreamSynchronize(c->stream));
        c->other_work = false;
    }
    const int dev = c->cfg.device;
    std::unique_lock<std::mutex> pair_lk(g_pair_mu[dev < AZ_MAX_DEVICES ? dev : 0]);
    // A pair wants every CU: a second pair on the GPU at the same time would leave neither resident.  The pair launches
    // of the contexts of this process on one GPU take turns, and it is the host that waits for the other context's
    // launch to finish: a wait on the GPU (hipStreamWaitEvent) leaves blocked barrier packets at the heads of this
    // context's hardware queues for as long as the other pair runs, and a blocked packet on a queue of higher
    // priority stops the command processor from serving a queue that shares its pipe -- when the tree kernel's
    // stream was a high-priority one that was the other context's own stream, which then never signalled the event.
    // The streams have one priority now; the tree kernel's stream still waits for no event of the context's stream:
    // the gate kernel in front of the tree kernel waits in memory for the tower workgroups instead.
    if (dev < AZ_MAX_DEVICES) {
        if (!g_pair_done[dev]) HIPCHK(hipEventCreateWithFlags(&g_pair_done[dev], hipEventDisableTiming));
        else if (g_pair_last[dev] != c) HIPCHK(hipEventSynchronize(g_pair_done[dev]));
        g_pair_last[dev] = c;
    }
    const bool sampled = c->prof_every > 0 && c->prof_used + 2 <= c->prof_pool.size();
    if (sampled) { c->pair_spans.push_back({c->prof_used, rounds}); (void)hipEventRecord(c->prof_pool[c->prof_used], c->stream); }
    // round 1 of the launch evaluates what the last select left in leaf_planes: the tower reads planes_wt only
    HIPCHK(hipMemcpy2DAsync(c->pair.planes_wt, (size_t)pr.plane_stride, c->d.leaf_planes, (size_t)3 * c->A, (size_t)3 * c->A,
                            (size_t)c->cfg.G, hipMemcpyDeviceToDevice, c->stream));
    const int blocks = az_launch_tower_pair(c->stream, c->prec, (const uint8_t*)c->pair.planes_wt, c->stem_wbf, c->stem_b, c->conv_wf, c->conv_b,
                                            c->head_pt, c->d.policy, c->d.value, c->cfg.G, c->cfg.N, c->Gd, 2 * c->num_res,
                                            c->d_sat, &pr, c->n_cu);
    if (blocks <= 0) return fail(AZ_E_STATE, "az_selfplay_run_pair: the persistent tower kernel does not fit");
    if (sampled) { (void)hipEventRecord(c->prof_pool[c->prof_used + 1], c->stream); c->prof_used += 2; }
    c->pair_started += (unsigned)blocks;
    pr.started_want = c->pair_started;
    if (!(dbg & 4)) az_launch_round_pair(c->stream2, &c->d, &pr, AZ_MODE_SELFPLAY, std::min(c->cfg.G, c->n_cu));
    // the context's stream goes on behind both kernels
    HIPCHK(hipEventRecord(c->pair_join, c->stream2));
    HIPCHK(hipStreamWaitEvent(c->stream, c->pair_join, 0));
    if (dev < AZ_MAX_DEVICES) HIPCHK(hipEventRecord(g_pair_done[dev], c->stream));
    pair_lk.unlock();
    HIPCHK(hipGetLastError());
    c->pair_base += (unsigned)rounds;
    c->rounds += rounds;
    c->pair_launches++; c->pair_rounds += rounds;
    return AZ_OK;
}

extern "C" int az_pair_read(az_ctx* c, double* launch_avg_us, int64_t* launches, int64_t* rounds, double* tree_step_avg_us) {
    NEED(c); NEED_GAMES(c); NEED(launch_avg_us); NEED(launches); NEED(rounds); NEED(tree_step_avg_us);
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    AZCHK(pair_check(c));
    double us = 0.0; int64_t n = 0, r = 0;
    for (auto& sp : c->pair_spans) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->prof_pool[sp.first], c->prof_pool[sp.first + 1]));
        us += ms * 1000.0; n++; r += sp.second;
    }
    *launch_avg_us = n ? us / n : 0.0; *launches = n; *rounds = r;
    // ticks (100 MHz) the tree waves worked, waits excluded, over every game-step of every pair launch so far
    std::vector<unsigned long long> busy(c->cfg.G);
    HIPCHK(hipMemcpy(busy.data(), c->pair.busy, sizeof(unsigned long long) * c->cfg.G, hipMemcpyDeviceToHost));
    double ticks = 0.0;
    for (auto b : busy) ticks += (double)b;
    const double steps = (double)c->pair_rounds * c->cfg.G;
    *tree_step_avg_us = steps > 0 ? ticks / steps * 0.01 : 0.0;
    return AZ_OK;
}

extern "C" int az_selfplay_run(az_ctx* c, int rounds) {
    NEED(c); NEED_GAMES(c);
    if (c->mode != AZ_MODE_SELFPLAY) return fail(AZ_E_STATE, "az_selfplay_begin has not been called");
    {   // AZ_PAIR=1: every self-play round of a context that can, as the persistent pair (A/B runs, soak tests)
        const char* e = getenv("AZ_PAIR");
        if (e && atoi(e) != 0 && !pair_unfit(c)) return az_selfplay_run_pair(c, rounds);
    }
    HIPCHK(hipSetDevice(c->cfg.device));
    for (int r = 0; r < rounds; ++r) {
        AZCHK(run_evaluator(c));
        const bool on = prof_begin(c, 1, c->prof_every > 0 && (c->rounds % c->prof_every) == 0, c->stream);
        az_launch_round(c->stream, &c->d, 3, AZ_MODE_SELFPLAY, 0, c->cfg.G);
        prof_end(c, on, c->stream);
        // an empty span right behind a sampled one: what two event records cost on this stream when nothing lies
        // between them, i.e. what every span above over-states its kernel by
        if (on) prof_end(c, prof_begin(c, 2, true, c->stream), c->stream);
        c->rounds++;
    }
    HIPCHK(hipGetLastError());
    return AZ_OK;
}

// Rows of ring r (not being appended to by anything that is still running) -> the caller's buffers; what does not
// fit stays, shifted to the front.  Blocking copies: they do not order against the context's stream.
static int drain_ring(az_ctx* c, int r, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward) {
    const az_ctx::Ring& R = c->rings[r];
    unsigned long long* ctr = c->d.counters + 4 + 2 * r;
    unsigned long long cnt = 0;
    HIPCHK(hipMemcpy(&cnt, ctr, 8, hipMemcpyDeviceToHost));
    const size_t A = c->A;
    const int n = (int)std::min<unsigned long long>(cnt, (unsigned long long)cap);
    if (n > 0) {
        HIPCHK(hipMemcpy(cells, R.cells, (size_t)n * A, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(to_move, R.tomove, n, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(policy, R.policy, (size_t)n * A * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(reward, R.reward, (size_t)n * 2 * 4, hipMemcpyDeviceToHost));
    }
    if (n > 0 && (unsigned long long)n < cnt) {
        // keep the tail: shift it to the front, n rows at a time from the low end so that the source
        // and destination of one copy never overlap
        const size_t rest = (size_t)(cnt - n);
        for (size_t off = 0; off < rest; off += (size_t)n) {
            const size_t k = std::min<size_t>((size_t)n, rest - off);
            HIPCHK(hipMemcpy(R.cells + off * A, R.cells + (off + n) * A, k * A, hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(R.tomove + off, R.tomove + off + n, k, hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(R.policy + off * A, R.policy + (off + n) * A, k * A * 4, hipMemcpyDeviceToDevice));
            HIPCHK(hipMemcpy(R.reward + off * 2, R.reward + (off + n) * 2, k * 8, hipMemcpyDeviceToDevice));
        }
    }
    cnt -= n;
    HIPCHK(hipMemcpy(ctr, &cnt, 8, hipMemcpyHostToDevice));
    return n;
}

// Everything finished so far, oldest first: waits for the context's stream, then empties the retired ring and
// the active one.
extern "C" int az_replay_drain(az_ctx* c, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward) {
    NEED(c); NEED_GAMES(c);
    if (cap < 0) return fail(AZ_E_ARG, "cap %d is negative", cap);
    if (cap > 0) { NEED(cells); NEED(to_move); NEED(policy); NEED(reward); }
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    AZCHK(pair_check(c));                                    // rows of a pair that fell out of step are nobody's games
    const size_t A = c->A;
    const int n0 = drain_ring(c, c->active ^ 1, cap, cells, to_move, policy, reward);
    if (n0 < 0 || n0 == cap) return n0;
    const int n1 = drain_ring(c, c->active, cap - n0, cells + (size_t)n0 * A, to_move + n0, policy + (size_t)n0 * A, reward + (size_t)n0 * 2);
    return n1 < 0 ? n1 : n0 + n1;
}

// The producer's overlap (generate.cpp:110 hands a finished game's rows over while the next search is already the
// reference's next statement; here the next MOVE of every game is already queued on the GPU): az_replay_flip retires
// the ring the launches queued so far append to -- launches queued after it append to the other ring -- and
// az_replay_drain_retired waits for exactly those earlier launches (an event recorded at the flip, not the whole
// stream) and hands over the retired ring's rows while the later launches keep the GPU busy.
extern "C" int az_replay_flip(az_ctx* c) {
    NEED(c); NEED_GAMES(c);
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipEventRecord(c->flip_ev, c->stream));
    c->active ^= 1;
    const az_ctx::Ring& R = c->rings[c->active];
    c->d.ring_cells = R.cells; c->d.ring_tomove = R.tomove; c->d.ring_policy = R.policy; c->d.ring_reward = R.reward;
    c->d.ring_ctr = c->d.counters + 4 + 2 * c->active;
    return AZ_OK;
}

extern "C" int az_replay_drain_retired(az_ctx* c, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward) {
    NEED(c); NEED_GAMES(c);
    if (cap < 0) return fail(AZ_E_ARG, "cap %d is negative", cap);
    if (cap > 0) { NEED(cells); NEED(to_move); NEED(policy); NEED(reward); }
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipEventSynchronize(c->flip_ev));                 // never recorded (no flip yet): returns at once, the ring is empty
    AZCHK(pair_check(c));                                    // rows of a pair that fell out of step are nobody's games
    return drain_ring(c, c->active ^ 1, cap, cells, to_move, policy, reward);
}

extern "C" int az_get_stats(az_ctx* c, az_stats* s) {
    NEED(c); NEED(s);
    HIPCHK(hipSetDevice(c->cfg.device));
    // one reduction kernel behind the queued work and one 256-byte copy
    unsigned long long k[32], sat = 0;
    memset(k, 0, sizeof k);
    if (!c->tower_only) {
        HIPCHK(hipMemsetAsync(c->d_stats + 26, 0, 6 * sizeof(unsigned long long), c->stream));   // the phase clocks are added into
        az_launch_stats(c->stream, &c->d, c->d_stats);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(k, c->d_stats, sizeof k, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipMemcpyAsync(&sat, c->d_sat, sizeof sat, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (!c->tower_only) AZCHK(pair_check(c));
    memset(s, 0, sizeof *s);
    s->saturated = (int64_t)sat;
    s->sims = (int64_t)k[0]; s->moves = (int64_t)k[2]; s->games = (int64_t)k[3];
    s->replay_rows = (int64_t)(k[4] + k[6]); s->replay_dropped = (int64_t)(k[5] + k[7]);
    s->rounds = c->rounds;
    s->evals = (int64_t)k[16]; s->descents = (int64_t)k[17]; s->tree_levels = (int64_t)k[18];
    s->children_scanned = (int64_t)k[19]; s->children_expanded = (int64_t)k[20];
    s->tree_restarts = (int32_t)k[21];
    // simulations backed up = those of completed searches + the progress of searches in flight
    s->sims_backed_up = (int64_t)(k[0] + k[22]);
    s->error_games = (int32_t)k[23];
    s->max_nodes = (int32_t)k[24];
    s->max_edges = (int32_t)k[25];
    for (int i = 0; i < 6; ++i) c->tree_ticks[i] = k[26 + i];
    return AZ_OK;
}

extern "C" int az_tree_timing(az_ctx* c, int on) {
    NEED(c); NEED_GAMES(c);
    c->d.timing = on ? 1 : 0;
    return AZ_OK;
}

extern "C" int az_tree_timing_read(az_ctx* c, double* select_us, double* step_us, double* append_us, double* backup_us,
                                   double* root_us, int64_t* steps) {
    NEED(c); NEED_GAMES(c); NEED(select_us); NEED(step_us); NEED(append_us); NEED(backup_us); NEED(root_us); NEED(steps);
    az_stats st;
    AZCHK(az_get_stats(c, &st));
    *select_us = c->tree_ticks[0] * 0.01; *step_us = c->tree_ticks[1] * 0.01; *append_us = c->tree_ticks[2] * 0.01;
    *backup_us = c->tree_ticks[3] * 0.01; *root_us = c->tree_ticks[4] * 0.01; *steps = (int64_t)c->tree_ticks[5];
    return AZ_OK;
}

// ---------------------------------------------------------------- profiling
extern "C" int az_profile_enable(az_ctx* c, int every) {
    NEED(c);
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->prof_every = every;
    c->prof_calls = 0;
    c->prof_used = 0;
    c->prof_spans.clear();
    c->pair_spans.clear();
    if (every > 0 && c->prof_pool.empty()) {
        c->prof_pool.reserve(8192);
        for (int i = 0; i < 8192; ++i) {
            hipEvent_t e;
            HIPCHK(hipEventCreate(&e));
            c->prof_pool.push_back(e);
        }
    }
    return AZ_OK;
}

extern "C" int az_profile_read(az_ctx* c, double* conv_avg_us, int64_t* conv_launches,
                               double* tree_avg_us, int64_t* tree_launches) {
    NEED(c); NEED(conv_avg_us); NEED(conv_launches); NEED(tree_avg_us); NEED(tree_launches);
    HIPCHK(hipSetDevice(c->cfg.device));
    HIPCHK(hipStreamSynchronize(c->stream));
    double sum[3] = {0, 0, 0};
    int64_t cnt[3] = {0, 0, 0};
    for (auto& sp : c->prof_spans) {
        float ms = 0.f;
        HIPCHK(hipEventElapsedTime(&ms, c->prof_pool[sp.first], c->prof_pool[sp.first + 1]));
        sum[sp.second] += ms * 1000.0;
        cnt[sp.second]++;
    }
    *conv_avg_us = cnt[0] ? sum[0] / cnt[0] : 0.0; *conv_launches = cnt[0];
    *tree_avg_us = cnt[1] ? sum[1] / cnt[1] : 0.0; *tree_launches = cnt[1];
    c->prof_empty_us = cnt[2] ? sum[2] / cnt[2] : 0.0; c->prof_empty_n = cnt[2];
    return AZ_OK;
}

extern "C" int az_profile_overhead(az_ctx* c, double* empty_span_us, int64_t* spans) {
    NEED(c); NEED(empty_span_us); NEED(spans);
    *empty_span_us = c->prof_empty_us; *spans = c->prof_empty_n;
    return AZ_OK;
}

// Host twin of the device rules (az_rules.h), exposed so that the CPU test-suite
// can check it against the oracle without a GPU.
extern "C" int az_host_env_step(int N, int K, uint8_t* cells, int* to_move, int action, float* reward) {
    if (N < 1 || K < 1 || !cells || !to_move || !reward) return fail(AZ_E_ARG, "null argument");
    return az_rules_step(N, K, cells, to_move, action, reward);
}

// replay.hpp:80-112 (host helper): rot90(r,{1,2}) then .flip(1), r = 0..3.
extern "C" int az_replay_symmetries(int N, const uint8_t* cells, int to_move, const float* policy,
                                    uint8_t* states_out, float* policies_out) {
    if (N < 1 || !cells || !policy || !states_out || !policies_out) return fail(AZ_E_ARG, "null argument");
    const int A = N * N;
    int v = 0;
    for (int r = 0; r < 4; ++r)
        for (int f = 0; f < 2; ++f, ++v)
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) {
                    const int ii = f ? N - 1 - i : i;
                    int sy, sx;
                    if (r == 0) { sy = ii; sx = j; }
                    else if (r == 1) { sy = j; sx = N - 1 - ii; }
                    else if (r == 2) { sy = N - 1 - ii; sx = N - 1 - j; }
                    else { sy = N - 1 - j; sx = ii; }
                    const uint8_t cv = cells[sy * N + sx];
                    uint8_t* so = states_out + (size_t)v * 3 * A;
                    so[i * N + j] = cv == 1;
                    so[A + i * N + j] = cv == 2;
                    so[2 * A + i * N + j] = (uint8_t)to_move;
                    policies_out[(size_t)v * A + i * N + j] = policy[sy * N + sx];
                }
    return AZ_OK;
}
