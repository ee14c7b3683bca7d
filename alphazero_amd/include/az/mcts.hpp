// mcts.hpp -- MCTS<Env> (mcts.hpp:18-581) over libazhip: same constructor,
// search_probs / load / clear, public attributes.  nthreads = 1 is the reference's
// single-thread search with bit-exact visit counts; nthreads = T > 1 runs T
// descents per round under virtual loss in a fixed order (the reference's T
// threads interleave nondeterministically).  batch_size is accepted for source
// compatibility: the leaf batch is whatever one round collects.  Throughput
// comes from many games in lockstep (az::SelfPlay in selfplay.hpp).
#pragma once
#include <chrono>
#include <iostream>
#include <vector>

#include "evaluator.hpp"

namespace az {

template <class Env>
class MCTS {
    friend class Evaluator<Env>;
public:
    using S = typename Env::state_type;
    using ID = typename Env::id_type;
    using A = typename Env::action_type;
    using P = FloatTensor;

    MCTS(int nthreads_, int batch_size_, float vl_, float c_puct_)      // mcts.hpp:41-73
        : nthreads(nthreads_), batch_size(batch_size_), vl(vl_), c_puct(c_puct_),
          evaluator(Evaluator<Env>::get(this, batch_size_)) {
        alpha = 10.f / (float)(env.get_board_size() * env.get_board_size());
    }

    // mcts.hpp:105-155
    P search_probs(const S& state_, int iter_budget_, int verbosity_ = 0) {
        verbosity = verbosity_;
        const auto start = std::chrono::system_clock::now();
        evaluator.reset_stat();
        const int N = env.get_board_size();
        std::vector<int32_t> hist((size_t)N * N, 0);
        int32_t len = (int32_t)state_.id.size();
        for (int i = 0; i < len; ++i) hist[(size_t)i] = state_.id[(size_t)i][0] * N + state_.id[(size_t)i][1];
        check(az_env_load(evaluator.ctx, hist.data(), &len));
        P actions_probs(env.get_action_shape());
        iter_budget = iter_budget_;
        check(az_search(evaluator.ctx, iter_budget_, actions_probs.data()));
        count = iter_budget_ > 1 ? iter_budget_ : 1;
        if (verbosity >= 1) {
            float avg; int nn;
            std::tie(avg, nn) = evaluator.retrieve_stat();
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - start);
            std::cout << "(LOG) threads: " << nthreads << " | iteration: " << count << " | average batch size: " << avg
                      << " | NN count: " << nn << " | time(ms): " << ms.count() << std::endl;
        }
        return actions_probs;
    }
    void load() { evaluator.setup(); }                                   // mcts.hpp:457-460
    void clear() { check(az_tree_clear(evaluator.ctx)); }                // mcts.hpp:462-465

    int nthreads;
    int batch_size;
    float vl;
    float c_puct;
    Env& env = Env::get();
    Evaluator<Env>& evaluator;
    int count = 0;
    int iter_budget = 0;
    float alpha;
    int verbosity = 0;
};

}  // namespace az
