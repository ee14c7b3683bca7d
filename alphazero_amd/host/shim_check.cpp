// shim_check.cpp -- exercises the reference-named C++ shims end to end on the
// GPU (used by tests/test_gpu_shim.py): MCTS<Env>::search_probs with root reuse
// over a short game, PVNetwork::forward, ReplayBuffer(GEN) framing.
#include <cstdio>
#include <fstream>

#include "../include/az/env.hpp"
#include "../include/az/mcts.hpp"
#include "../include/az/replay.hpp"

using Env = az::TicTacToe;

int main(int argc, char** argv) {
    const int budget = argc > 1 ? atoi(argv[1]) : 50;
    const int nthreads = argc > 2 ? atoi(argv[2]) : 1;
    { std::ofstream f("ckpt_location.txt"); f << "shim_0.pt"; }
    az::MCTS<Env> agent(nthreads, 16, 3, 3);
    Env& env = Env::get();
    agent.load();
    auto state = env.reset();
    agent.search_probs(state, budget, 0);                 // warm-up as generate.cpp:77
    bool done = false;
    int moves = 0;
    while (!done) {
        // the second move is searched at verbosity 3: the reference's phase report (mcts.hpp:545-575)
        az::FloatTensor policy = agent.search_probs(state, budget, moves == 1 ? 3 : 1);
        double sum = 0;
        for (float p : policy.v) sum += p;
        const int point = (int)az::argmax(policy);
        printf("move %d point %d policy_sum %.6f\n", moves, point, sum);
        az::FloatTensor reward;
        std::tie(state, reward, done) = env.step(state, {point / Env::N, point % Env::N});
        ++moves;
        if (done) printf("done reward %g %g\n", reward.v[0], reward.v[1]);
    }
    agent.clear();
    // PVNetwork shim
    az::PVNetwork net(3, az::NetConfig::get(2).resblocks(), 3, 1);
    std::vector<float> blob = az::fresh_blob(3, 3, az::NetConfig::get(2).resblocks(), 1);
    net->load_blob(blob.data(), blob.size());
    az::ByteTensor x({2, 3, 3, 3});
    x.v[4] = 1;
    az::FloatTensor p, v;
    std::tie(p, v) = net(x);
    double ps = 0;
    for (int i = 0; i < 9; ++i) ps += p.v[i];
    printf("pvnet policy_sum %.6f value_sum %.6f\n", ps, v.v[0] + v.v[1]);
    // Blokus Env shim: a random game through reset / possible_actions / step / get_player
    {
        az::Blokus& benv = az::Blokus::get();
        az::Blokus::state_type bs = benv.reset();
        bool bdone = false;
        az::FloatTensor brew;
        int bmoves = 0;
        unsigned rng = 12345u;
        while (!bdone) {
            const auto acts = benv.possible_actions(bs, benv.get_player(bs));
            if (bmoves == 0 && acts.size() != 58) { printf("blokus: %zu opening moves\n", acts.size()); return 1; }
            rng = rng * 1664525u + 1013904223u;
            std::tie(bs, brew, bdone) = benv.step(bs, acts[(rng >> 8) % acts.size()]);
            ++bmoves;
        }
        printf("blokus game %d moves reward %g %g %g %g board %lld\n", bmoves, brew.v[0], brew.v[1], brew.v[2], brew.v[3],
               (long long)benv.get_board(bs).numel());
    }
    printf("shim ok\n");
    return 0;
}
