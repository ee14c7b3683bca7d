/* san_driver.cpp -- a short tour of the oracle (tree search, self-play loop, tower, Blokus rules and search) for
 * the sanitizer build: `make -C oracle sanitize` compiles this file with az_oracle.cpp and blokus_oracle.cpp under
 * -fsanitize=address,undefined and runs it.  TEST INFRASTRUCTURE ONLY. */
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "az_oracle.h"

extern "C" {
void* azo_blokus_new(void);
void azo_blokus_free(void*);
int azo_blokus_legal(void*, uint8_t* mask);
int azo_blokus_random_move(void*, uint64_t seed, uint32_t stream, uint32_t counter, float* reward, int* done);
int azo_blokus_search(void*, int budget, float vl, float c_puct, const float* noise, int32_t* actions, int32_t* n,
                      float* v, float* p, int32_t* root_n, int64_t* n_nodes);
}

int main() {
    /* tree search with reuse across moves, multi-leaf rounds, a whole self-play game */
    const int N = 5, K = 4, A = N * N;
    azo_mcts* m = azo_mcts_create(N, K, 3.f, 3.f, azo_eval_hash, nullptr);
    std::vector<float> pol(A), noise(A, 1.f / A);
    std::vector<int32_t> hist;
    for (int mv = 0; mv < 6; ++mv) {
        if (azo_mcts_search(m, hist.data(), (int)hist.size(), 60, noise.data(), pol.data()) != 0) return 1;
        int best = 0;
        for (int i = 1; i < A; ++i) if (pol[i] > pol[best]) best = i;
        hist.push_back(best);
    }
    const int ne = azo_mcts_dump(m, 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    std::vector<int32_t> pa(ne), ac(ne), nn(ne), kd(ne);
    std::vector<float> vv(ne), pp(ne);
    azo_mcts_dump(m, ne, pa.data(), ac.data(), nn.data(), vv.data(), pp.data(), kd.data());
    azo_mcts_clear(m);
    if (azo_mcts_search_k(m, nullptr, 0, 77, 5, noise.data(), pol.data()) != 0) return 1;
    azo_mcts_clear(m);
    float avg = 25.f; int64_t gc = 0; uint32_t ctr = 0; azo_game_result res;
    std::vector<uint8_t> cells((size_t)A * A), tm(A);
    std::vector<float> rows((size_t)A * A);
    std::vector<int32_t> acts(A);
    azo_selfplay_game(m, 20, 3, 0, nullptr, &avg, &gc, &ctr, A, cells.data(), tm.data(), rows.data(), acts.data(), &res);
    azo_mcts_destroy(m);
    /* tower */
    const int Cs[3] = {8, 8, 8};
    const size_t nb = azo_tower_blob_size(N, 3, 1, Cs, 3);
    std::vector<float> blob(nb);
    for (size_t i = 0; i < nb; ++i) blob[i] = 0.01f * (float)((i * 2654435761u) % 97) - 0.3f;
    for (size_t i = 0; i < nb; ++i) if (blob[i] < 0.05f && blob[i] > -0.05f) blob[i] = 0.07f;
    azo_tower* t = azo_tower_create(N, 3, 1, Cs, 3, blob.data(), nb);
    if (!t) return 1;
    std::vector<uint8_t> boards((size_t)2 * 3 * A, 0);
    boards[3] = 1; boards[A + 7] = 1;
    std::vector<float> p2((size_t)2 * A), v2(4);
    azo_tower_forward(t, boards.data(), 2, p2.data(), v2.data());
    azo_tower_destroy(t);
    /* symmetries */
    std::vector<uint8_t> so((size_t)8 * 3 * A);
    std::vector<float> po((size_t)8 * A);
    azo_replay_symmetries(N, 3, boards.data(), p2.data(), so.data(), po.data());
    /* Blokus: a few random moves, the legal set, a search */
    void* b = azo_blokus_new();
    float rw[4]; int done = 0;
    for (int i = 0; i < 12 && !done; ++i) azo_blokus_random_move(b, 5, 1, (uint32_t)i, rw, &done);
    std::vector<uint8_t> mask(91 * 400);
    const int k = azo_blokus_legal(b, mask.data());
    std::vector<int32_t> ba(k), bn(k);
    std::vector<float> bv(k), bp(k);
    int32_t rn = 0; int64_t nodes = 0;
    if (k > 0) azo_blokus_search(b, 30, 3.f, 3.f, nullptr, ba.data(), bn.data(), bv.data(), bp.data(), &rn, &nodes);
    azo_blokus_free(b);
    std::printf("oracle sanitizer tour ok (%d edges, %d blokus actions)\n", ne, k);
    return 0;
}
