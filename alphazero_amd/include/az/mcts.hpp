// mcts.hpp -- MCTS<Env> (mcts.hpp:18-581) over libazhip: same constructor,
// search_probs / load / clear, public attributes.  nthreads = 1 is the reference's
// single-thread search with bit-exact visit counts; nthreads = T > 1 runs T
// descents per round under virtual loss in a fixed order (the reference's T
// threads interleave nondeterministically).  batch_size is accepted for source
// compatibility: the leaf batch is whatever one round collects.  Throughput
// comes from many games in lockstep (az::SelfPlay in selfplay.hpp).
#pragma once
#include <algorithm>
#include <chrono>
#include <iomanip>
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
        double t0[5] = {0, 0, 0, 0, 0};
        int64_t steps0 = 0;
        if (verbosity >= 3) {                      // __log_init (mcts.hpp:514-519): the phase clocks run on the GPU
            check(az_tree_timing(evaluator.ctx, 1));
            check(az_tree_timing_read(evaluator.ctx, &t0[0], &t0[1], &t0[2], &t0[3], &t0[4], &steps0));
            check(az_profile_enable(evaluator.ctx, 1));
        }
        check(az_search(evaluator.ctx, iter_budget_, actions_probs.data()));
        count = iter_budget_ > 1 ? iter_budget_ : 1;
        if (verbosity >= 3) __log_v3(t0, steps0);
        if (verbosity >= 1) {
            float avg; int nn;
            std::tie(avg, nn) = evaluator.retrieve_stat();
            const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - start);
            std::cout << "(LOG) threads: " << nthreads << " | iteration: " << count << " | average batch size: " << avg
                      << " | NN count: " << nn << " | time(ms): " << ms.count() << std::endl;
        }
        return actions_probs;
    }
    // mcts.hpp:545-575, same lines, milliseconds.  Select / Step / Append / Backup / Root Prune are read off the GPU's
    // clock inside the tree kernel at the reference's own phase boundaries; Net is the duration of the evaluator's
    // launches (HIP events); Eval = Step + Net + Append as in mcts.hpp:248-303.
    void __log_v3(const double (&t0)[5], int64_t steps0) {
        double t[5];
        int64_t steps = 0;
        check(az_tree_timing_read(evaluator.ctx, &t[0], &t[1], &t[2], &t[3], &t[4], &steps));
        double net_us = 0, tree_us = 0;
        int64_t net_count = 0, tree_n = 0;
        check(az_profile_read(evaluator.ctx, &net_us, &net_count, &tree_us, &tree_n));
        check(az_profile_enable(evaluator.ctx, 0));
        check(az_tree_timing(evaluator.ctx, 0));
        const double select_f = (t[0] - t0[0]) * 1e-3, step_f = (t[1] - t0[1]) * 1e-3, append_f = (t[2] - t0[2]) * 1e-3;
        const double backup_f = (t[3] - t0[3]) * 1e-3, create_f = (t[4] - t0[4]) * 1e-3;
        const double net_f = net_us * (double)net_count * 1e-3, eval_f = step_f + net_f + append_f;
        const int64_t step_count = std::max<int64_t>(steps - steps0, 1), nets = std::max<int64_t>(net_count, 1);
        const int cnt = std::max(count, 1);
        std::cout << "(Root Prune) Total: " << std::setw(8) << create_f << std::endl;
        std::cout << "(Select)     Total: " << std::setw(8) << select_f << " | Avg: " << select_f / cnt << std::endl;
        std::cout << "(Eval)       Total: " << std::setw(8) << eval_f << " | Avg: " << eval_f / cnt << std::endl;
        std::cout << "    (Step)   Total: " << std::setw(8) << step_f << " | Avg: " << step_f / (double)step_count << std::endl;
        std::cout << "    (Net)    Total: " << std::setw(8) << net_f << " | Avg: " << net_f / (double)nets << std::endl;
        std::cout << "    (Append) Total: " << std::setw(8) << append_f << " | Avg: " << append_f / (double)nets << std::endl;
        std::cout << "(Backup)     Total: " << std::setw(8) << backup_f << " | Avg: " << backup_f / cnt << std::endl;
        std::cout << "Simulation count: " << count << std::endl;
        std::cout << "Step count: " << step_count << std::endl;
        std::cout << "Net count: " << net_count << std::endl;
        std::cout << std::endl;
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
