/*
 * az_oracle.h -- CPU restatement of the reference self-play hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped HIP path includes, links
 * or calls this library; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and only as the checker.
 *
 * Every function cites the reference file:line it restates
 * (paths relative to the reference checkout: src/mcts.hpp, src/node.hpp, ...).
 *
 * Pinning status:
 *   - tree search (mcts.hpp/node.hpp): pinned against the compiled, unmodified
 *     reference headers (oracle/_ref/az_ref, built by oracle/Makefile) through
 *     the fixtures under tests/golden/.
 *   - tower (network.hpp): pinned against the compiled reference network.hpp
 *     the same way.
 *   - environments: PARITY UNPINNED.  The reference's env sources are git
 *     submodules that are empty in the checkout (.gitmodules:1-9); the rules
 *     here follow the call-site contract (mcts.hpp:251-271, generate.cpp:75-103)
 *     and standard m,n,k-game rules.
 */
#ifndef AZ_ORACLE_H
#define AZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- env ---- */
/* cells: N*N bytes, 0 empty, 1 stone of player 0, 2 stone of player 1.       */
void azo_env_reset(int N, uint8_t* cells, int* to_move);
/* returns 1 done, 0 not done, -1 illegal action.  reward is f32[2] indexed by
 * absolute player id (mcts.hpp:336 `result[player]`). */
int  azo_env_step(int N, int K, uint8_t* cells, int* to_move, int action, float reward[2]);
/* u8 planes [3,N,N]: stones of player 0, stones of player 1, side to move.    */
void azo_env_planes(int N, const uint8_t* cells, int to_move, uint8_t* planes);
/* legal actions (empty cells) in row-major order; returns the count.          */
int  azo_env_legal(int N, const uint8_t* cells, int32_t* actions);

/* ---------------------------------------------------------- evaluators ---- */
typedef void (*azo_eval_fn)(void* user, int N, const uint8_t* planes,
                            float* policy /*N*N*/, float* value /*2*/);
/* Integer-hash pseudo network: exactly reproducible on CPU and GPU. */
void azo_eval_hash(void* user, int N, const uint8_t* planes, float* policy, float* value);

/* Table evaluator: answers from recorded (planes -> policy,value) pairs and
 * fails (sets *miss) on a board it has never seen. */
typedef struct azo_table azo_table;
azo_table* azo_table_create(int N);
void azo_table_destroy(azo_table*);
void azo_table_add(azo_table*, const uint8_t* planes, const float* policy, const float* value);
int  azo_table_misses(const azo_table*);
void azo_eval_table(void* table, int N, const uint8_t* planes, float* policy, float* value);

/* --------------------------------------------------------------- tower ---- */
/* fp32 restatement of PVNetwork::forward (network.hpp:117-123), eval mode.
 * blob layout: see azo_tower_blob_size(). */
typedef struct azo_tower azo_tower;
size_t azo_tower_blob_size(int N, int c_in, int c_out, const int* Cs, int n_Cs);
azo_tower* azo_tower_create(int N, int c_in, int c_out, const int* Cs, int n_Cs,
                            const float* blob, size_t n_floats);
void azo_tower_destroy(azo_tower*);
/* boards u8 [B,c_in,N,N] -> policy f32 [B,c_out*N*N] (softmax probs), value f32 [B,2] */
void azo_tower_forward(const azo_tower*, const uint8_t* boards, int B, float* policy, float* value);
void azo_eval_tower(void* tower, int N, const uint8_t* planes, float* policy, float* value);

/* ---------------------------------------------------------------- mcts ---- */
typedef struct azo_mcts azo_mcts;
azo_mcts* azo_mcts_create(int N, int K, float vl, float c_puct, azo_eval_fn fn, void* user);
void azo_mcts_destroy(azo_mcts*);
/* One MCTS::search_probs call (mcts.hpp:105-155) at nthreads=1.  The state is
 * given as the move history since reset.  noise: Dirichlet noise for a fresh
 * root in legal-action order (mcts.hpp:432-446), or NULL to draw it with
 * azo_dirichlet from the instance's generator.  out_policy f32 [N*N].
 * returns 0, or -1 on an illegal history. */
int  azo_mcts_search(azo_mcts*, const int32_t* history, int hist_len, int budget,
                     const float* noise, float* out_policy);
/* The same search with up to K descents per round (virtual loss between them,
 * evaluations batched, backups in descent order): the defined-order form of the
 * reference's nthreads = K (mcts.hpp:54-72,216-231).  K = 1 == azo_mcts_search. */
int  azo_mcts_search_k(azo_mcts*, const int32_t* history, int hist_len, int budget, int K,
                       const float* noise, float* out_policy);
/* Make every later search of this instance (azo_selfplay_game included) use K
 * descents per round. */
void azo_mcts_set_leaves(azo_mcts*, int K);
void azo_mcts_clear(azo_mcts*);                 /* mcts.hpp:462-465 */
void azo_mcts_seed(azo_mcts*, uint64_t seed);
int  azo_mcts_root_n(const azo_mcts*);
long azo_mcts_eval_count(const azo_mcts*);
/* Canonical dump of the whole tree in BFS order from the root, children in
 * action order.  One record per edge (= child stub, node.hpp:21-22):
 *   parent (BFS index of the expanded parent), action, n, v, p, kind
 *   kind: 0 stub (never visited), 1 expanded, 2 terminal p0 wins,
 *         3 terminal p1 wins, 4 terminal draw.
 * returns the number of edges (call with cap=0 to size). */
int  azo_mcts_dump(const azo_mcts*, int cap, int32_t* parent, int32_t* action,
                   int32_t* n, float* v, float* p, int32_t* kind);

/* util.h:42-58 restated with an explicit generator state */
void azo_dirichlet(uint64_t* rng_state, int size, float alpha, float* out);

/* ------------------------------------------------------------ self-play ---- */
/* Counter-based generator shared bit-for-bit with the HIP path. */
uint32_t azo_philox_u32(uint64_t seed, uint32_t stream, uint32_t counter, uint32_t lane);

typedef struct {
    int   n_moves;
    float reward[2];
} azo_game_result;

/* One game of generate.cpp:72-120 for game slot `slot`: warm-up search, per
 * move search -> record -> sample/argmax -> step.  rows are appended to
 * cells_out [max_moves, N*N] u8, tomove_out [max_moves], policy_out
 * [max_moves, N*N] f32, actions_out [max_moves].  avg_len / game_count are
 * the running statistics of generate.cpp:25-29 and are updated in place. */
int azo_selfplay_game(azo_mcts*, int budget, uint64_t seed, uint32_t slot,
                      const float* noise, float* avg_len, int64_t* game_count,
                      uint32_t* sample_counter,
                      int max_moves, uint8_t* cells_out, uint8_t* tomove_out,
                      float* policy_out, int32_t* actions_out, azo_game_result* res);

/* replay.hpp:62-114: the 8 symmetry variants of one (state, policy) row in
 * emission order.  states_out u8 [8,C,N,N]; policies_out f32 [8,N,N]. */
void azo_replay_symmetries(int N, int C, const uint8_t* state, const float* policy,
                           uint8_t* states_out, float* policies_out);

#ifdef __cplusplus
}
#endif
#endif
