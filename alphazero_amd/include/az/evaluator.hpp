// evaluator.hpp + mcts.hpp shims share one device context per Env: in the
// reference the Evaluator singleton owns the network and MCTS owns the tree
// (evaluator.hpp:39-77, mcts.hpp:478-481); here both live in one az_ctx whose
// leaf batch is the set of lockstep games.
#pragma once
#include <cstdlib>
#include <iostream>
#include <string>
#include <tuple>

#include "../../../include/az_hip.h"
#include "netconfig.hpp"
#include "network.hpp"
#include "util.hpp"

namespace az {

template <class Env> class MCTS;

template <class Env>
class Evaluator {
    friend class MCTS<Env>;
public:
    // evaluator.hpp:39-43 (singleton => one MCTS per Env per process)
    static Evaluator& get(MCTS<Env>* mcts, int batch_size) {
        static Evaluator singleton(mcts, batch_size);
        return singleton;
    }
    // evaluator.hpp:137-148
    void setup() {
        WeightFile w;
        bool changed = false;
        const std::string path = load_network(w, model_path, changed, [this](WeightFile& f) {
            f.N = Env::N; f.c_in = 3; f.c_out = 1; f.Cs = NetConfig::get(2).resblocks();
            f.blob = fresh_blob(Env::N, 3, f.Cs, 0);
        });
        if (changed) {
            if (w.N != Env::N || w.Cs != NetConfig::get(2).resblocks()) throw Error("checkpoint shape does not match the environment");
            check(az_load_weights(ctx, w.blob.data(), w.blob.size()));
            model_path = path;
        }
    }
    // evaluator.hpp:150-165: running mean batch size and NN call count of one search
    void reset_stat() { az_stats s; check(az_get_stats(ctx, &s)); evals0 = s.evals; rounds0 = s.rounds; }
    std::tuple<float, int> retrieve_stat() {
        az_stats s; check(az_get_stats(ctx, &s));
        const long calls = (long)(s.rounds - rounds0);
        return {calls ? (float)(s.evals - evals0) / (float)calls : 0.f, (int)calls};
    }
    az_ctx* context() const { return ctx; }
    ~Evaluator() { az_ctx_destroy(ctx); }

private:
    Evaluator(MCTS<Env>* mcts_, int batch_size_) : mcts(mcts_), batch_size(batch_size_) {
        NetConfig& netconf = NetConfig::get(2);                     // evaluator.hpp:57
        std::cout << netconf.channels_to_string() << std::endl;
        const std::vector<int> Cs = netconf.resblocks();
        az_config cfg{};
        const char* dev = getenv("AZ_DEVICE");
        const char* mb = getenv("AZ_MAX_BUDGET");
        cfg.device = dev ? atoi(dev) : 0;
        cfg.N = Env::N; cfg.K = Env::K; cfg.G = 1;
        cfg.n_Cs = (int)Cs.size();
        for (size_t i = 0; i < Cs.size(); ++i) cfg.Cs[i] = Cs[i];
        cfg.vl = mcts_->vl; cfg.c_puct = mcts_->c_puct;
        // nthreads searcher threads (mcts.hpp:54-72) become nthreads descents per round
        cfg.leaves_per_round = mcts_->nthreads < 1 ? 1 : (mcts_->nthreads > 64 ? 64 : mcts_->nthreads);
        cfg.max_budget = mb ? atoi(mb) : 1600;
        cfg.eval_mode = AZ_EVAL_TOWER;
        cfg.noise_mode = AZ_NOISE_DEVICE;
        cfg.seed = std::random_device{}();
        check(az_ctx_create(&ctx, &cfg));
        setup();
        std::cout << "* --------------- Evaluator --------------- *\n"
                  << "Model: " << model_path << "\nDevice: MI355X (HIP)\nBatch size: lockstep games\n"
                  << "* ----------------------------------------- *" << std::endl;
    }
    MCTS<Env>* mcts;
    int batch_size;
    std::string model_path;
    az_ctx* ctx = nullptr;
    int64_t evals0 = 0, rounds0 = 0;
};

}  // namespace az
