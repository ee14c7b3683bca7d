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
    // The games run as `contexts` engine contexts on their own HIP streams (default 2, or
    // AZ_CONTEXTS): a tree block fits on a CU beside a resident tower workgroup, so one
    // context's tree round executes under the other's tower kernel.
    SelfPlay(int games, int iter_budget, float vl = 3, float c_puct = 3, int device = 0, uint64_t seed = 0,
             int slot_base = 0, int contexts = 0) : G(games), budget(iter_budget) {
        const std::vector<int> Cs = NetConfig::get(2).resblocks();
        if (contexts <= 0) { const char* e = getenv("AZ_CONTEXTS"); contexts = e ? atoi(e) : 2; }
        if (contexts < 1 || games < 2 * contexts) contexts = 1;
        int first = 0;
        for (int i = 0; i < contexts; ++i) {
            const int n = games / contexts + (i < games % contexts ? 1 : 0);
            az_config cfg{};
            cfg.device = device; cfg.N = Env::N; cfg.K = Env::K; cfg.G = n;
            cfg.n_Cs = (int)Cs.size();
            for (size_t k = 0; k < Cs.size(); ++k) cfg.Cs[k] = Cs[k];
            cfg.vl = vl; cfg.c_puct = c_puct; cfg.max_budget = iter_budget;
            cfg.eval_mode = AZ_EVAL_TOWER; cfg.noise_mode = AZ_NOISE_DEVICE;
            cfg.seed = seed; cfg.game_slot_base = slot_base + first;
            az_ctx* c = nullptr;
            check(az_ctx_create(&c, &cfg));
            ctxs.push_back(c);
            first += n;
        }
    }
    ~SelfPlay() { for (az_ctx* c : ctxs) az_ctx_destroy(c); }
    SelfPlay(const SelfPlay&) = delete;
    SelfPlay& operator=(const SelfPlay&) = delete;
    // agent.load() (generate.cpp:73): re-read ckpt_location.txt, reload on change
    bool load() {
        WeightFile w;
        bool changed = false;
        const std::string path = load_network(w, model_path, changed, [this](WeightFile& f) {
            f.N = Env::N; f.c_in = 3; f.c_out = 1; f.Cs = NetConfig::get(2).resblocks();
            f.blob = fresh_blob(Env::N, 3, f.Cs, 0);
        });
        if (changed) {
            for (az_ctx* c : ctxs) check(az_load_weights(c, w.blob.data(), w.blob.size()));
            model_path = path;
        }
        return changed;
    }
    void begin() { for (az_ctx* c : ctxs) check(az_selfplay_begin(c, budget)); }
    // launches are asynchronous: the contexts are fed in slices so that every stream stays busy
    void run(int rounds) {
        for (int done = 0; done < rounds;) {
            const int k = rounds - done < 16 ? rounds - done : 16;
            for (az_ctx* c : ctxs) check(az_selfplay_run(c, k));
            done += k;
        }
    }
    // finished rows -> sink(board u8 [3,N,N], policy f32 [1,N,N], reward f32 [2])
    int drain(const std::function<void(const ByteTensor&, const FloatTensor&, const FloatTensor&)>& sink) {
        const int N = Env::N, A = N * N, cap = 4096;
        std::vector<uint8_t> cells((size_t)cap * A), tm(cap);
        std::vector<float> pol((size_t)cap * A), rew((size_t)cap * 2);
        int total = 0;
        for (az_ctx* c : ctxs)
            for (;;) {
                const int n = az_replay_drain(c, cap, cells.data(), tm.data(), pol.data(), rew.data());
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
    // counters summed over the contexts (high-water marks: maximum)
    az_stats stats() {
        az_stats t{};
        for (az_ctx* c : ctxs) {
            az_stats s;
            check(az_get_stats(c, &s));
            t.sims += s.sims; t.evals += s.evals; t.moves += s.moves; t.games += s.games;
            t.replay_rows += s.replay_rows; t.replay_dropped += s.replay_dropped;
            t.error_games += s.error_games; t.tree_restarts += s.tree_restarts; t.sims_backed_up += s.sims_backed_up;
            t.descents += s.descents; t.tree_levels += s.tree_levels;
            t.children_scanned += s.children_scanned; t.children_expanded += s.children_expanded;
            t.rounds = s.rounds > t.rounds ? s.rounds : t.rounds;
            t.max_nodes = s.max_nodes > t.max_nodes ? s.max_nodes : t.max_nodes;
            t.max_edges = s.max_edges > t.max_edges ? s.max_edges : t.max_edges;
        }
        return t;
    }
    az_ctx* context(int i = 0) const { return ctxs[(size_t)i]; }
    int contexts() const { return (int)ctxs.size(); }
    const int G, budget;
private:
    std::vector<az_ctx*> ctxs;
    std::string model_path;
};

}  // namespace az
