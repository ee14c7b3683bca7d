/*
 * az_hip.h -- C ABI of libazhip.so: the MI355X-native self-play hot path
 * (env step -> PUCT select with virtual loss -> ResNet leaf evaluation ->
 * expand -> backup) behind the reference's Env / MCTS / Evaluator / PVNetwork
 * interfaces.  The reference has no FFI of its own; its boundary is the C++
 * class API, so each entry point cites the reference member it replaces
 * (paths relative to the reference checkout).
 *
 * Conventions: every function returns 0 on success or a negative AZ_E_* code
 * and records a message readable with az_last_error() (the reference throws
 * std::runtime_error: node.hpp:50-53, mcts.hpp:374-375, util.h:69-82).  No
 * exception crosses this boundary.  The caller owns every host buffer; the
 * library owns all device memory.  One context per GPU; a context is not
 * thread-safe, distinct contexts may be driven from distinct host threads.
 * G games advance in lockstep: one round = one simulation per game
 * (= MCTS::__simulate_once at nthreads=1, mcts.hpp:183-208) with the G leaf
 * boards evaluated as one batch (the Evaluator's role, evaluator.hpp:80-124).
 */
#ifndef AZ_HIP_H
#define AZ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZ_OK            0
#define AZ_E_ARG        -1   /* bad argument */
#define AZ_E_HIP        -2   /* HIP runtime error (message has the call) */
#define AZ_E_CAPACITY   -3   /* a capacity fixed at az_ctx_create is too small for the request */
#define AZ_E_STATE      -4   /* call not valid in the current state */
#define AZ_E_NOWEIGHTS  -5   /* tower used before az_load_weights */

#define AZ_MAX_LAYERS 64

/* eval_mode */
#define AZ_EVAL_TOWER    0   /* PVNetwork forward on the device (network.hpp) */
#define AZ_EVAL_HASH     1   /* integer-hash pseudo network, bit-reproducible on CPU */
#define AZ_EVAL_EXTERNAL 2   /* caller evaluates leaves (az_search_select/apply) */
/* noise_mode */
#define AZ_NOISE_EXPLICIT 0  /* Dirichlet noise from az_set_noise */
#define AZ_NOISE_DEVICE   1  /* drawn on the device, alpha = 0.05 (util.h:42) */
/* operand format of the tower's 3x3 convolutions (fp32 accumulate, fp32 residual stream, fp32 heads) */
#define AZ_PREC_FP16     0   /* fp16 operands (11-bit significand), saturating: the default and the throughput mode.
                                Within 1e-3 of the fp32 network (network.hpp:117-123) on random-init towers, also with
                                sharpened heads, and on towers with trained-network-like BatchNorm and flat heads; a
                                tower that is both (BatchNorm renormalising every layer, peaked policy) amplifies the
                                operand rounding to 2e-3 .. 6e-3 on the policy */
#define AZ_PREC_BF16     1   /* bf16 operands: same MFMA rate in cycles, ~4 % faster in wall time (lower power),
                                8x coarser rounding: within 1e-3 only on flat (random-init) outputs */
#define AZ_PREC_FP16X3   2   /* compensated fp16, the parity mode for deep or peaked towers: every operand is a pair
                                hi + lo of fp16 values (22-bit significand) and every 3x3 convolution runs as three
                                MFMA passes (hi*hi + hi*lo + lo*hi) into the fp32 accumulation, one launch per pass on
                                the per-layer kernels: fp32-class agreement with network.hpp at about a third of the
                                per-layer path's rate */

typedef struct az_ctx az_ctx;

typedef struct az_config {
    int   device;                 /* HIP device ordinal */
    int   N;                      /* board side: Env::get_board_size (mcts.hpp:51) */
    int   K;                      /* stones in a row that win (3 tictactoe, 5 gomoku) */
    int   G;                      /* concurrent games on this GPU */
    int   n_Cs;                   /* length of Cs; num_res = n_Cs-1 (network.hpp:105) */
    int   Cs[AZ_MAX_LAYERS];      /* NetConfig::resblocks() (netconfig.hpp:19-22) */
    float vl;                     /* MCTS ctor vl_ (mcts.hpp:41) */
    float c_puct;                 /* MCTS ctor c_puct_; truncated to int by Node::ucb (node.hpp:47) */
    int   node_cap;               /* expanded nodes per tree arena, 0 = 4*max_budget+64; a tree that outgrows it is
                                     dropped and its search restarts from a fresh root (az_stats.tree_restarts) */
    int   max_budget;             /* largest iter_budget that will be requested */
    int   eval_mode;
    int   noise_mode;
    int   precision;
    uint64_t seed;                /* counter-based generator seed (sampling, device noise) */
    int   game_slot_base;         /* global index of this context's game 0 (multi-GPU sharding) */
    int   leaves_per_round;       /* L: descents per tree per round, 0 = 1.  L = 1 is the reference at
                                     nthreads = 1; L > 1 is its nthreads = L (mcts.hpp:54-72,216-231) in one
                                     defined order: descents in sequence under virtual loss, one batched
                                     evaluation of the G*L slots, backups in descent order; a descent that
                                     meets a stub already awaiting evaluation is undone and ends the round */
    /* PVNetwork(board, Cs, in, out) (network.hpp:104): input planes and policy channels of the tower, 0 = the
     * m,n,k environments' 3 and 1; value_dim = entries of the value head, 0 = the reference's 2
     * (network.hpp:73,81).  Other values (Blokus: 5, 91, 4) give a context whose tower is used through
     * az_tower_forward*; its m,n,k search entry points return AZ_E_STATE. */
    int   c_in, c_out, value_dim;
    /* What az_load_weights does with a blob whose BatchNorm-folded weights leave fp16's range (one beyond +-65504, or
     * an output channel whose largest weight is nonzero and below 2^-14) when the context runs fp16 operands: 0 =
     * refuse it with AZ_E_ARG and keep the weights in use; 1 = take it with bf16 operands (fp32's range, 8-bit
     * significand) and say so in az_weights_info.precision_in_use. */
    int   range_fallback;
} az_config;

/* What the last az_load_weights* call on the context found and chose. */
typedef struct az_weights_info {
    int     precision_in_use;     /* AZ_PREC_* the tower's kernels run now */
    int     fell_back;            /* 1: the blob left fp16's range and az_config.range_fallback put the context on bf16 */
    int64_t overflow_weights;     /* folded 3x3 weights beyond +-65504 */
    int64_t underflow_channels;   /* output channels whose largest folded weight is nonzero and below 2^-14 */
    float   max_abs_weight;       /* largest folded 3x3 weight */
    float   min_channel_max;      /* smallest nonzero per-channel maximum */
} az_weights_info;

typedef struct az_stats {
    int64_t sims;                 /* completed simulations (MCTS::count, mcts.hpp:170) */
    int64_t evals;                /* leaf/root boards submitted to the evaluator (evaluator.hpp:150-165) */
    int64_t rounds;               /* lockstep rounds executed */
    int64_t moves;                /* self-play moves played */
    int64_t games;                /* self-play games finished (generate.cpp:115) */
    int64_t replay_rows;          /* rows currently waiting in the replay ring */
    int64_t replay_dropped;       /* rows dropped because the ring was full */
    int32_t max_nodes;            /* high-water mark of expanded nodes in any tree */
    int32_t max_edges;
    int32_t error_games;          /* games stopped by an error (no selectable child, mcts.hpp:374-375) */
    int32_t tree_restarts;        /* searches restarted from a fresh root because their arena was full (the
                                     reference's heap has no cap; it does the same on a find() miss, mcts.hpp:392-414) */
    int64_t sims_backed_up;       /* simulations backed up so far, searches in flight included */
    /* what the tree kernel's memory traffic is made of (summed over games since the context's
     * creation or its last az_selfplay_begin): */
    int64_t descents;             /* root-to-leaf descents that stood (__select_and_eval, mcts.hpp:211-237) */
    int64_t tree_levels;          /* tree levels those descents walked (sum of path lengths) */
    int64_t children_scanned;     /* child edges read by __choose (mcts.hpp:350-381) on the way */
    int64_t children_expanded;    /* child edges written by _append_children (mcts.hpp:425-455) */
    int64_t saturated;            /* tower tiles (one launch per evaluation) or boards (one launch per convolution) whose
                                     residual stream reached the largest fp16 value, 65504, at the heads: the stream was
                                     clamped on the way and those answers are not the network's.  0 with bf16 operands. */
} az_stats;

const char* az_last_error(void);
/* torch::cuda::is_available / device choice of the Evaluator (evaluator.hpp:141-144): GPUs this process sees */
int az_device_count(void);

/* MCTS<Env>::MCTS + Evaluator<Env>::Evaluator (mcts.hpp:41-73, evaluator.hpp:48-77):
 * allocates all device state for G games and the tower. */
int az_ctx_create(az_ctx** out, const az_config* cfg);
int az_ctx_destroy(az_ctx*);                       /* ~MCTS / ~Evaluator (mcts.hpp:80-87) */
int az_sync(az_ctx*);                              /* wait for the context's stream */
void* az_stream(az_ctx*);                          /* the hipStream_t all work is issued on */

/* load_network (util.h:60-105) without the file: flat fp32 blob in module
 * registration order (see alphazero_amd/synth.py blob_layout); BatchNorm is
 * folded here.  Host pointer, or a device pointer with the _dev variant
 * (used after an RCCL broadcast). */
size_t az_weights_size(const az_ctx*);
int az_load_weights(az_ctx*, const float* blob, size_t n_floats);
int az_load_weights_dev(az_ctx*, const float* blob_dev, size_t n_floats);
/* No counterpart in the reference (fp32 throughout): where the last blob's folded weights sit in fp16's range and
 * the operand format the context runs as a result; also filled after a refused load. */
int az_weights_info_get(const az_ctx*, az_weights_info*);

/* Env::reset (generate.cpp:75), Env::step (mcts.hpp:251,261; generate.cpp:103),
 * Env::get_board (mcts.hpp:271), Env::possible_actions (mcts.hpp:427), batched
 * over the G games.  actions[g] = y*N+x, or -1 to leave game g untouched.
 * reward f32 [G,2] by absolute player, done u8 [G]. */
int az_env_reset(az_ctx*);
int az_env_step(az_ctx*, const int32_t* actions, float* reward, uint8_t* done);
int az_env_boards(az_ctx*, uint8_t* planes /* [G,3,N,N] */);
int az_env_legal_mask(az_ctx*, uint8_t* mask /* [G,N,N] */);
int az_env_set(az_ctx*, const uint8_t* cells /* [G,N*N] 0/1/2 */, const uint8_t* to_move /* [G] */);
/* Env::get_id (mcts.hpp:388): set every game from its move history, hist i32
 * [G,N*N] (row-major, y*N+x), hist_len i32 [G]; keeps the trees so that find()
 * (node.hpp:83-102) can reuse them. */
int az_env_load(az_ctx*, const int32_t* hist, const int32_t* hist_len);

/* PVNetwork::forward (network.hpp:117-123) in eval mode as driven by
 * Evaluator::run (evaluator.hpp:113-117): boards u8 [B,3,N,N] ->
 * policy f32 [B,c_out*N*N] (softmax probabilities), value f32 [B,value_dim]. */
int az_tower_forward(az_ctx*, const uint8_t* boards, int B, float* policy, float* value);
int az_tower_forward_dev(az_ctx*, const uint8_t* boards_dev, int B, float* policy_dev, float* value_dev);
/* The same forward for a caller that will read only some policy entries -- the search reads a leaf's priors at its
 * legal actions (_append_children, mcts.hpp:425-455), a few hundred of Blokus' 36 400 -- wanted_dev u32
 * [B, c_out * N]: bit x of word co * N + y set = entry (co, y, x) is wanted.  Those entries get the bits
 * az_tower_forward_dev gives them (the softmax still runs over every logit); the others are left untouched. */
int az_tower_forward_masked_dev(az_ctx*, const uint8_t* boards_dev, int B, float* policy_dev, float* value_dev,
                                const uint32_t* wanted_dev);
/* PVNetwork's constructor arguments as the context holds them (network.hpp:104): board side, `in`, `out`,
 * value entries, and the largest batch one call takes (any pointer may be NULL) */
int az_tower_shape(const az_ctx*, int* N, int* c_in, int* c_out, int* value_dim, int* max_batch);

/* No counterpart in the reference.  az_tower_forward_dev for callers that cannot allocate device memory
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc): on a context
 * whose tower runs one launch per convolution the activation images live in a 64 MiB array that the
 * library holds once per GPU (file scope, until the process ends) instead of in buffers sized by
 * az_config.G, and the call makes no allocation of any kind.  B is bounded by that array
 * (az_tower_static_max_batch; larger batches return AZ_E_CAPACITY), not by az_config.G.  Every call on a
 * GPU shares the GPU's array: two such calls that overlap in time on one GPU race. */
int az_tower_forward_static(az_ctx*, const uint8_t* boards_dev, int B, float* policy_dev, float* value_dev);
size_t az_tower_static_max_batch(const az_ctx*);

/* util.h:42-58 made an input: noise f32 [G,N*N], used in legal-action order
 * for every freshly created root (mcts.hpp:432-446). */
int az_set_noise(az_ctx*, const float* noise);
/* The noise each game's latest fresh root was mixed with, f32 [G,N*N], entry k = k-th legal action; with
 * AZ_NOISE_DEVICE that is the device's own dirichlet(n_actions, 0.05) draw (util.h:42-58). */
int az_get_noise(az_ctx*, float* noise);

/* MCTS::search_probs (mcts.hpp:105-155) for all G games from their current
 * env states, including root reuse (mcts.hpp:385-421): out f32 [G,N*N]. */
int az_search(az_ctx*, int iter_budget, float* out_policy);
/* The same search with the evaluator outside: begin, then alternate
 * select (planes u8 [G*L,3,N,N]; pending i32 [G*L]: 0 nothing, 1 root, 2 leaf,
 * 3 terminal; slot g*L+k is descent k of game g) and apply (policy f32
 * [G*L,N*N], value f32 [G*L,2]) until select returns *n_running == 0, then
 * read the result. */
int az_search_begin(az_ctx*, int iter_budget);
int az_search_select(az_ctx*, uint8_t* planes, int32_t* pending, int* n_running);
int az_search_apply(az_ctx*, const float* policy, const float* value);
int az_search_result(az_ctx*, float* out_policy, int32_t* root_n /* [G] or NULL */);
int az_tree_clear(az_ctx*);                        /* MCTS::clear (mcts.hpp:462-465) */

/* Parity introspection: the arena of game g.  edges are {p,n,v,child}; child
 * >= 0 expanded node id (nodes numbered in BFS order, node 0 is the root; node i owns edges
 * [node_first[i], node_first[i] + node_count[i])), -1 stub, -2/-3/-4 terminal (p0 wins /
 * p1 wins / draw).  Call with cap 0 to size. */
int az_tree_dump(az_ctx*, int game, int node_cap, int edge_cap, int32_t* n_nodes, int32_t* n_edges,
                 int32_t* node_first, int32_t* node_count,
                 float* edge_p, int32_t* edge_n, float* edge_v, int32_t* edge_child, int32_t* edge_action);

/* generate.cpp:72-120 on the device for all G slots: runs `rounds` lockstep
 * rounds of self-play (search -> record row -> sample/argmax -> step -> on
 * game end attach rewards, restart).  Rows accumulate in a device ring. */
int az_selfplay_begin(az_ctx*, int iter_budget);
int az_selfplay_run(az_ctx*, int rounds);
/* A simulation whose descent ends in a terminal leaf with a stored reward calls no network (mcts.hpp:232-233), but
 * in a round of az_selfplay_run* every game owns one evaluation, used or not.  With max_chained > 0 a game runs up to
 * that many such simulations back to back inside one round (backup, next select) until a leaf that needs the
 * evaluator is out, so the evaluation of the round is not spent on a stale board.  The order of every tree's
 * simulations, and with it every search result, move and row, is unchanged; what changes is how many simulations a
 * round holds (az_stats.sims_backed_up grows faster than rounds x games).  0 (the default) = one simulation per game
 * per round.  Needs leaves_per_round 1.  In lockstep rounds the tree launch lasts as long as its longest chain.
 * AZ_CHAIN=k in the environment sets it at every az_selfplay_begin (for host programs that do not call this). */
int az_selfplay_chain(az_ctx*, int max_chained);
/* The same `rounds` rounds (generate.cpp:72-120; per search mcts.hpp:105-155) as ONE launch of two persistent
 * kernels that run side by side: the tower, one workgroup per CU walking its boards round after round, and the tree
 * waves, one per CU beside it.  A game's next evaluation starts as soon as that game's select has published its
 * planes, and its next select as soon as its evaluation is there: the tree work of one game runs under the tower
 * work of another and no launch boundary separates rounds.  Games share no tree state, so every game's trees, rows
 * and counters are those of az_selfplay_run bit for bit (rows of different games may reach the ring in another
 * order).  Needs the one-launch tower (az_tower_is_fused), leaves_per_round 1 and the GPU to itself: the two kernels
 * wait for each other in memory, so both must be resident at once.  If they are not (another process or context on
 * the GPU, a profiler that serialises launches), a wait runs out after AZ_PAIR_TIMEOUT_MS (default 4000), both
 * kernels leave, and az_sync / az_get_stats / az_pair_read return AZ_E_STATE from then on.
 * The error text names the wait that ran out (tower workgroup or tree wave, game, the word it wanted and saw).
 * The call returns once the launch is queued, like az_selfplay_run, with one exception: when the last pair launch on
 * this GPU was another context's of this process, it first waits on the host for that launch to end (two pairs at
 * once would leave neither resident; see api.hip for why the wait is not left to the GPU), and when evaluations or
 * lockstep rounds of this context are still running it waits for those (the tree waves of a launch are resident at
 * once and give the tower kernel one time-out to begin).  The two kernels run on two streams of the context, of
 * the same priority and on different hardware queues (tried out when the context is made: a process whose other
 * streams leave no such queue gets AZ_E_STATE here and plays lockstep rounds); work of OTHER streams that shares
 * those queues and runs for seconds can still hold one of the two kernels up until the time-out.
 * With AZ_PAIR=1 in the environment az_selfplay_run takes this path whenever the context can. */
int az_selfplay_run_pair(az_ctx*, int rounds);
/* What the pair launches since az_profile_enable / az_selfplay_begin took: mean duration of the sampled tower
 * launches (HIP events on the context's stream), how many were sampled and the rounds they ran, and the mean time a
 * tree wave worked on one game's step (100 MHz clock inside the kernel, waits excluded; since az_selfplay_begin). */
int az_pair_read(az_ctx*, double* launch_avg_us, int64_t* launches, int64_t* rounds, double* tree_step_avg_us);
/* Drain finished rows: cells u8 [n,N*N] (0/1/2), to_move u8 [n], policy f32
 * [n,N*N], reward f32 [n,2] (replay.hpp:56-60,62-79 before symmetry expansion).
 * Returns the number of rows copied (<= cap) or a negative error. */
int az_replay_drain(az_ctx*, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward);
/* The hand-over of finished rows without stopping the GPU (generate.cpp:110-113 while the next searches are already
 * queued): a context has two rings.  az_replay_flip retires the ring that the launches queued SO FAR append to;
 * launches queued afterwards append to the other one.  az_replay_drain_retired waits for those earlier launches
 * only (an event recorded at the flip) and drains the retired ring, same row layout and return value as
 * az_replay_drain, while the later launches run.  az_replay_drain itself waits for the whole stream and empties
 * both rings, retired first. */
int az_replay_flip(az_ctx*);
int az_replay_drain_retired(az_ctx*, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward);
int az_get_stats(az_ctx*, az_stats*);

/* generate.cpp:37-123 as ONE process over every GPU of the node (the reference runs one generator process;
 * its MCTS/Evaluator pair, mcts.hpp:41-73, becomes one context per GPU): games are sharded by GPU, trees
 * never interact, so nothing in the data path is a collective.  cfg describes ONE context (cfg->G games
 * each; cfg->device is ignored, game_slot_base numbers the group's games from cfg->game_slot_base); the
 * group holds contexts_per_device contexts on each of the n_devices GPUs, one host thread per GPU issues that
 * GPU's launches.  az_group_load_weights is load_network (util.h:60-105) for the node: the blob goes to the
 * first GPU once, RCCL broadcasts it over xGMI (ncclBroadcast on a communicator of the group's GPUs, bound
 * from librccl.so.1 at run time), every GPU folds and packs its copy on the device.  az_group_replay_drain is
 * the gather of finished rows to the process that owns the PUSH socket (replay.hpp:41-54): rows of every
 * context in context order (GPU-major), same layout as az_replay_drain.  Counters of az_group_get_stats are
 * sums over the contexts (rounds, max_nodes, max_edges: maxima). */
typedef struct az_group az_group;
int az_group_create(az_group** out, const az_config* cfg, const int* devices, int n_devices, int contexts_per_device);
int az_group_destroy(az_group*);
int az_group_contexts(const az_group*);
az_ctx* az_group_context(az_group*, int i);
int az_group_load_weights(az_group*, const float* blob, size_t n_floats);
int az_group_selfplay_begin(az_group*, int iter_budget);
int az_group_selfplay_run(az_group*, int rounds);
int az_group_sync(az_group*);
int az_group_replay_drain(az_group*, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward);
int az_group_replay_flip(az_group*);              /* az_replay_flip on every context */
int az_group_replay_drain_retired(az_group*, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward);
int az_group_get_stats(az_group*, az_stats*);
/* The host-side merge az_group_replay_drain is made of, usable on its own (and without a GPU): sources are
 * drained in order through fn(src, cap_left, row buffers at the next free row) until cap rows are out. */
typedef int (*az_drain_fn)(void* src, int cap, uint8_t* cells, uint8_t* to_move, float* policy, float* reward);
int az_merge_drain(int n_src, az_drain_fn fn, void* const* srcs, int A, int cap,
                   uint8_t* cells, uint8_t* to_move, float* policy, float* reward);
int az_ctx_cells(const az_ctx*);                  /* N*N of the context's board */
int az_ctx_device(const az_ctx*);                 /* the HIP device ordinal the context lives on (az_config.device) */

/* ---- Blokus: 20x20, four players, 21 polyominoes each (rules stated in alphazero_amd/csrc/az_blokus_rules.h).
 * The reference names a Blokus environment (.gitmodules:7-9) without shipping it and its MCTS cannot drive
 * four players (replay.hpp:127, mcts.hpp:431), so these are the Env members of the call sites
 * (mcts.hpp:251-271,427; generate.cpp:75,103) for G lockstep games, on the rules this library defines:
 * action = o*400 + y*20 + x (orientation o of 91, bounding-box corner at row y, column x), policy layout
 * [91,20,20], planes u8 [5,20,20] (squares of players 0..3, player to move), reward f32 [4] by player. */
typedef struct az_blokus az_blokus;
int az_blokus_create(az_blokus** out, int device, int G, uint64_t seed, int game_slot_base);
int az_blokus_destroy(az_blokus*);
int az_blokus_reset(az_blokus*);                                                 /* Env::reset */
/* Env::step: actions i32 [G] (-1 = leave the game alone) -> reward f32 [G,4], done u8 [G] */
int az_blokus_step(az_blokus*, const int32_t* actions, float* reward, uint8_t* done);
/* Env::possible_actions of the player to move: mask u8 [G,91,20,20] (or NULL), count i32 [G] (or NULL) */
int az_blokus_legal_mask(az_blokus*, uint8_t* mask, int32_t* count);
int az_blokus_boards(az_blokus*, uint8_t* planes /* [G,5,20,20] */);             /* Env::get_board */
/* Env::get_player and bookkeeping (any pointer may be NULL): to_move u8 [G], over u8 [G], pieces placed
 * i32 [G], pieces-played masks u32 [G,4], final reward f32 [G,4] */
int az_blokus_state(az_blokus*, uint8_t* to_move, uint8_t* over, int32_t* moves, uint32_t* used, float* final_reward);
/* `steps` uniformly random legal moves per unfinished game, on the device (asynchronous): the k-th legal
 * action in index order, k = (u * count) >> 32, u = Philox4x32-10(seed, stream game_slot_base + g,
 * counter = pieces placed) word 0.  The env-only workload of the measurements. */
int az_blokus_playout(az_blokus*, int steps);
int az_blokus_sync(az_blokus*);
void* az_blokus_stream(az_blokus*);
/* MCTS over the Blokus games (MCTS::MCTS + search_probs, mcts.hpp:41-73,105-155, at nthreads = 1): the
 * reference's search arithmetic on four players and an irregular action set -- an edge is valued for the
 * player who chooses it (result[parent->player], mcts.hpp:336, with four entries), a node carries its own
 * child count, children are in action order, ties go to the first child.  Every search starts from a fresh
 * root unless az_blokus_search_reuse switched the reference's tree reuse on; a tree that outgrows its arena (edge_cap slots per game, 0 = (max_budget + 2) * 1024) is dropped and
 * searched again, and if it outgrows it a second time the search ends with the simulations it has (both counted
 * by az_blokus_search_stats' restarts).  az_blokus_search runs `budget` simulations per game with the integer-hash evaluator
 * (bit-reproducible on the CPU); begin / round / eval_io let an evaluator outside the library take the hash
 * evaluator's place: after a round its input planes are in *planes_dev, before the next round its outputs
 * must be in *policy_dev / *value_dev (softmax over all 91*400 actions, value per player). */
int az_blokus_search_config(az_blokus*, float vl, float c_puct, int max_budget, int edge_cap);
/* __make_root + find (mcts.hpp:385-421, node.hpp:83-102), to be called BEFORE az_blokus_search_config (it sizes the
 * arenas: two per game, default edge_cap (2 * max_budget + 64) * 1024): every search first looks its position up in
 * the game's kept tree by the moves played since the tree's root.  Found below the root: that subtree becomes the
 * tree (copied to the other arena, its root to slot 0) with its statistics and no Dirichlet noise; the root itself:
 * the visits keep accumulating; not found, or a child that was never expanded (undefined in the reference): a
 * fresh root.  az_blokus_tree_clear is MCTS::clear (mcts.hpp:462-465).  With reuse on, az_blokus_selfplay_moves is the
 * reference's loop in full: a warm-up search at every game start whose result is discarded (generate.cpp:77; it takes
 * one of the n_moves steps of that game), the tree carried from move to move, cleared at the game's end
 * (generate.cpp:113); a row then holds up to az_blokus_selfplay_row_cap() = 4 * max_budget policy entries. */
int az_blokus_search_reuse(az_blokus*, int on);
/* Before az_blokus_search_config: `leaves` (1..16) descents per game and round, the counterpart of MCTS(nthreads, ...)'s
 * searcher threads on one tree (mcts.hpp:54-72,161-174, virtual loss mcts.hpp:225-227) and of az_config.leaves_per_round
 * for the m,n,k games: a round evaluates az_blokus_eval_slots() = G * leaves positions (the evaluator's buffers and the
 * tower's batch are that large; slot g * leaves + k), applies them in slot order and issues up to `leaves` descents per
 * game, never more than the simulations still owed.  A descent that reaches a leaf an earlier descent of the round
 * is waiting on is taken back exactly and ends that game's descents for the round (the reference's thread would
 * block on the node's lock, mcts.hpp:224). */
int az_blokus_search_leaves(az_blokus*, int leaves);
int az_blokus_eval_slots(const az_blokus*);
int az_blokus_tree_clear(az_blokus*);
int az_blokus_selfplay_row_cap(const az_blokus*);
int az_blokus_set_noise(az_blokus*, const float* noise /* [G,n] or NULL = no noise */, int n);
int az_blokus_search(az_blokus*, int budget);
/* the same search with the leaves evaluated by `tower`, a context created with N = 20, c_in = 5, c_out = 91,
 * value_dim = 4 and G >= az_blokus_eval_slots(), on the same GPU (the Evaluator + PVNetwork pair of
 * evaluator.hpp:80-124 / network.hpp:117-123 for this Env): nothing leaves the device between the rounds */
int az_blokus_search_tower(az_blokus*, az_ctx* tower, int budget);
int az_blokus_search_begin(az_blokus*, int budget);
int az_blokus_search_round(az_blokus*, int* n_running /* or NULL */);
int az_blokus_eval_hash(az_blokus*);
int az_blokus_eval_io(az_blokus*, uint8_t** planes_dev, float** policy_dev, float** value_dev);
/* root children of game g in action order: actions, n, v, p; cap 0 = size only */
int az_blokus_search_result(az_blokus*, int game, int cap, int32_t* n_children, int32_t* root_n,
                            int32_t* actions, int32_t* n, float* v, float* p);
int az_blokus_search_stats(az_blokus*, int32_t* max_nodes, int32_t* max_slots, int32_t* restarts, int64_t* rounds);
/* dirichlet(n_actions, 0.05) of util.h:42-58 drawn on the device at every fresh root (on = 0: no noise); the
 * reference seeds its gamma draws from random_device, so only the distribution is defined */
int az_blokus_noise_device(az_blokus*, int on);
/* The move loop of generate.cpp:37-123 for the G games.  Per move: a search of `budget` simulations from a fresh
 * root (tower = NULL: the integer-hash evaluator), the row {planes, to move, search policy} buffered
 * (generate.cpp:85), the move chosen -- sampled in proportion to the visits while the 1-based move number is below
 * 0.2 x the running mean game length (initially 25), the first arg-max afterwards (generate.cpp:87-100) -- and
 * played; a finished game hands its rows to a ring of `ring_rows` rows with the final reward (generate.cpp:110)
 * and the next game starts in its slot.  Differences from the reference's loop, which cannot run this game: tree
 * reuse between moves and the warm-up search only with az_blokus_search_reuse (else every root is fresh),
 * four-entry rewards, and a row's policy is
 * kept sparse: entry j < counts[r] says policy[actions[j]] = probs[j] (= n_child / n_root, mcts.hpp:136-146) for
 * the visited root children in action order, every other entry of the [91,20,20] policy being zero. */
int az_blokus_selfplay_config(az_blokus*, int ring_rows);
int az_blokus_selfplay_moves(az_blokus*, az_ctx* tower /* or NULL */, int budget, int n_moves);
/* up to cap finished rows, oldest first, removed from the ring: planes u8 [cap,5,20,20], to_move u8 [cap],
 * counts i32 [cap], actions i32 / probs f32 [cap,max_budget] (-1 / 0 beyond counts), reward f32 [cap,4] */
int az_blokus_replay_drain(az_blokus*, int cap, uint8_t* planes, uint8_t* to_move, int32_t* counts,
                           int32_t* actions, float* probs, float* reward, int32_t* n_rows);
int az_blokus_selfplay_stats(az_blokus*, int64_t* moves, int64_t* games, int64_t* rows, int64_t* dropped);
/* the orientation table: piece, bounding box, cells (row, column) row-major; returns the number of cells */
int az_blokus_orient(int o, int* piece, int* h, int* w, int8_t* cells_yx /* [5][2] */);
/* the same rules on the host, one game in caller memory of az_blokus_host_state_size() bytes (C++ Env shim,
 * CPU tests): step returns 1 done, 0 running, AZ_E_ARG illegal; legal returns the count */
size_t az_blokus_host_state_size(void);
int az_blokus_host_reset(void* state);
int az_blokus_host_step(void* state, int action, float* reward /* [4] */);
int az_blokus_host_legal(const void* state, uint8_t* mask /* [91*400] or NULL */);
int az_blokus_host_info(const void* state, uint8_t* cells /* [400] */, int* to_move, int* over, uint32_t* used /* [4] */, int32_t* scores /* [4] */);

/* Env::step on the host (the rules the C++ Env shim and az_env_load use; same
 * rules as the device kernels): returns 1 done, 0 running, AZ_E_ARG illegal. */
int az_host_env_step(int N, int K, uint8_t* cells, int* to_move, int action, float* reward);

/* Measurement hook (no counterpart in the reference, whose only timers are the
 * verbosity-3 chrono buckets of mcts.hpp:187-201): HIP events on the context's
 * stream around every 3x3-convolution launch and every tree-round launch of
 * each `every`-th evaluator call; every = 0 switches sampling off. */
int az_profile_enable(az_ctx*, int every);
/* 1 when the tower runs as ONE launch per evaluation (az_tower_fused_kernel:
 * C <= 128 and (N+1)^2 <= 256), 0 when it runs one launch per convolution. */
int az_tower_is_fused(const az_ctx*);
/* Boards per evaluator launch during az_selfplay_run: G * leaves_per_round. */
int az_selfplay_batch(const az_ctx*);
/* The reference's verbosity-3 report (mcts.hpp:187-201,248-303,545-575) times its phases with std::chrono around
 * the calls; here the tree kernels read the GPU's 100 MHz clock at the same boundaries while az_tree_timing is on and
 * add the spans per game: select = the descent of __select_and_eval, step = Env::step at the leaf + handing the
 * board to the evaluator, append = _append_children, backup = __backup, root = __make_root (find + prune, and the
 * first expansion of a fresh root).  az_tree_timing_read returns the sums over all games since the counters were
 * last cleared (az_selfplay_begin; az_ctx_create), in microseconds, and the number of leaf steps.  The network's own
 * bucket is the evaluator launches' duration (az_profile_enable / az_profile_read). */
int az_tree_timing(az_ctx*, int on);
int az_tree_timing_read(az_ctx*, double* select_us, double* step_us, double* append_us, double* backup_us,
                        double* root_us, int64_t* steps);
int az_profile_read(az_ctx*, double* conv_avg_us, int64_t* conv_launches,
                    double* tree_avg_us, int64_t* tree_launches);
/* What a span over-states its kernel by: the mean duration of the empty spans (two event records with nothing
 * between them) taken right behind the sampled tree rounds, as of the last az_profile_read. */
int az_profile_overhead(az_ctx*, double* empty_span_us, int64_t* spans);

/* replay.hpp:80-112: the 8 dihedral variants of one row in emission order,
 * states u8 [8,3,N,N], policies f32 [8,N,N] (host-side helper). */
int az_replay_symmetries(int N, const uint8_t* cells, int to_move, const float* policy,
                         uint8_t* states_out, float* policies_out);

#ifdef __cplusplus
}
#endif
#endif /* AZ_HIP_H */
