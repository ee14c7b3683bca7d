// selfplay.hpp -- the many-game driver: G slots, each slot an independent
// instance of the reference's generate.cpp:72-120 loop, advanced in lockstep on
// one GPU.  Rows come back as (board, policy, reward) exactly as the reference
// buffers them (replay.hpp:56-60) before symmetry expansion.
#pragma once
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#include "../../../include/az_hip.h"
#include "../../csrc/az_rules.h"
#include "netconfig.hpp"
#include "network.hpp"
#include "util.hpp"

namespace az {

template <class Env>
class SelfPlay {
public:
    SelfPlay(int games, int iter_budget, float vl = 3, float c_puct = 3, int device = 0, uint64_t seed = 0,
             int slot_base = 0) : G(games), budget(iter_budget) {
        const std::vector<int> Cs = NetConfig::get(2).resblocks();
        az_config cfg{};
        cfg.device = device; cfg.N = Env::N; cfg.K = Env::K; cfg.G = games;
        cfg.n_Cs = (int)Cs.size();
        for (size_t i = 0; i < Cs.size(); ++i) cfg.Cs[i] = Cs[i];
        cfg.vl = vl; cfg.c_puct = c_puct; cfg.max_budget = iter_budget;
        cfg.eval_mode = AZ_EVAL_TOWER; cfg.noise_mode = AZ_NOISE_DEVICE;
        cfg.seed = seed; cfg.game_slot_base = slot_base;
        check(az_ctx_create(&ctx, &cfg));
    }
    ~SelfPlay() { az_ctx_destroy(ctx); }
    // agent.load() (generate.cpp:73): re-read ckpt_location.txt, reload on change
    bool load() {
        WeightFile w;
        bool changed = false;
        const std::string path = load_network(w, model_path, changed, [this](WeightFile& f) {
            f.N = Env::N; f.c_in = 3; f.c_out = 1; f.Cs = NetConfig::get(2).resblocks();
            f.blob = random_blob(az_weights_size(ctx), 0);
        });
        if (changed) {
            check(az_load_weights(ctx, w.blob.data(), w.blob.size()));
            model_path = path;
        }
        return changed;
    }
    void begin() { check(az_selfplay_begin(ctx, budget)); }
    void run(int rounds) { check(az_selfplay_run(ctx, rounds)); }
    // finished rows -> sink(board u8 [3,N,N], policy f32 [1,N,N], reward f32 [2])
    int drain(const std::function<void(const ByteTensor&, const FloatTensor&, const FloatTensor&)>& sink) {
        const int N = Env::N, A = N * N, cap = 4096;
        std::vector<uint8_t> cells((size_t)cap * A), tm(cap);
        std::vector<float> pol((size_t)cap * A), rew((size_t)cap * 2);
        int total = 0;
        for (;;) {
            const int n = az_replay_drain(ctx, cap, cells.data(), tm.data(), pol.data(), rew.data());
            check(n);
            for (int r = 0; r < n; ++r) {
                ByteTensor board({3, N, N});
                az_rules_planes(N, cells.data() + (size_t)r * A, tm[r], board.data());
                FloatTensor policy({1, N, N}), reward({2});
                std::copy(pol.begin() + (size_t)r * A, pol.begin() + (size_t)(r + 1) * A, policy.v.begin());
                reward.v[0] = rew[2 * (size_t)r]; reward.v[1] = rew[2 * (size_t)r + 1];
                sink(board, policy, reward);
            }
            total += n;
            if (n < cap) break;
        }
        return total;
    }
    az_stats stats() { az_stats s; check(az_get_stats(ctx, &s)); return s; }
    az_ctx* context() const { return ctx; }
    const int G, budget;
private:
    az_ctx* ctx = nullptr;
    std::string model_path;
};

}  // namespace az
