// selfplay.hpp -- the many-game driver: G slots per GPU, each slot an independent
// instance of the reference's generate.cpp:72-120 loop, advanced in lockstep on
// every GPU the process was given (az_group_*: one host thread per GPU, RCCL for the weights).  Rows come back as (board, policy, reward) exactly as the reference
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

// "AZ_GPUS": unset or "1" -> device (default 0) only; "all" -> every visible GPU; "n" -> GPUs 0..n-1;
// "a,b,c" -> exactly those.  n_visible < 0 means "not known": counts are taken at face value.
inline std::vector<int> parse_gpus(const char* spec, int fallback_device, int n_visible = -1) {
    std::vector<int> d;
    if (!spec || !*spec) return {fallback_device};
    const std::string s(spec);
    if (s == "all") {
        if (n_visible < 1) throw Error("AZ_GPUS=all needs the number of visible GPUs");
        for (int i = 0; i < n_visible; ++i) d.push_back(i);
        return d;
    }
    if (s.find(',') == std::string::npos) {
        const int n = atoi(s.c_str());
        if (n < 1) throw Error("AZ_GPUS: expected a count, a comma-separated list, or 'all'");
        for (int i = 0; i < n; ++i) d.push_back(i);
        return d;
    }
    size_t pos = 0;
    while (pos <= s.size()) {
        const size_t q = s.find(',', pos);
        const std::string tok = s.substr(pos, q == std::string::npos ? std::string::npos : q - pos);
        if (tok.empty()) throw Error("AZ_GPUS: empty entry in the device list");
        d.push_back(atoi(tok.c_str()));
        if (q == std::string::npos) break;
        pos = q + 1;
    }
    return d;
}

// "AZ_NET_CHANNELS": the tower's channel list as NetConfig::resblocks() would give it, "128x10" (ten entries of 128 =
// NetConfig 2, the default when unset) or "256x21" (BASELINE's 20-block tower), or a comma-separated list.  The
// reference's generator has NetConfig::get(2) compiled in (generate.cpp:58); this is how a trainer that was built with
// another table is followed without recompiling.
inline std::vector<int> parse_channels(const char* spec) {
    if (!spec || !*spec) return NetConfig::get(2).resblocks();
    const std::string s(spec);
    std::vector<int> c;
    const size_t x = s.find('x');
    if (x != std::string::npos) {
        const int ch = atoi(s.substr(0, x).c_str()), n = atoi(s.substr(x + 1).c_str());
        if (ch < 1 || n < 2 || n > 64) throw Error("AZ_NET_CHANNELS: expected <channels>x<entries>, entries 2..64");
        c.assign((size_t)n, ch);
        return c;
    }
    size_t pos = 0;
    while (pos <= s.size()) {
        const size_t q = s.find(',', pos);
        const int ch = atoi(s.substr(pos, q == std::string::npos ? std::string::npos : q - pos).c_str());
        if (ch < 1) throw Error("AZ_NET_CHANNELS: channel counts must be positive");
        c.push_back(ch);
        if (q == std::string::npos) break;
        pos = q + 1;
    }
    if (c.size() < 2 || c.size() > 64) throw Error("AZ_NET_CHANNELS: 2..64 entries");
    return c;
}

// "AZ_PRECISION": the operand format of the tower's kernels (az_config.precision): "fp16" (the default when unset),
// "bf16", or "fp16x3", the compensated mode that holds the reference's fp32 outputs to 1e-4 on every tower at about
// a quarter of the rate.
inline int parse_precision(const char* spec) {
    if (!spec || !*spec) return AZ_PREC_FP16;
    const std::string s(spec);
    if (s == "fp16") return AZ_PREC_FP16;
    if (s == "bf16") return AZ_PREC_BF16;
    if (s == "fp16x3") return AZ_PREC_FP16X3;
    throw Error("AZ_PRECISION: expected fp16, bf16 or fp16x3");
}

template <class Env>
class SelfPlay {
public:
    // `games` slots PER GPU on each GPU of `devices`, as `contexts` engine contexts per GPU (default 1, or
    // AZ_CONTEXTS).  One host thread per GPU issues the launches; RCCL carries the weights from the first GPU to
    // the others (az_group_load_weights).  With AZ_PAIR=1 in the environment run() is one launch of the persistent
    // tower / tree pair per call on every context that can (az_selfplay_run_pair), AZ_CHAIN=k lets simulations
    // that end in a stored terminal leaf ride along (az_selfplay_chain).
    // `channels`: the tower (NetConfig::resblocks() form); empty = NetConfig::get(2), the reference's (generate.cpp:58).
    // `precision`: AZ_PREC_* (see parse_precision).
    SelfPlay(int games, int iter_budget, float vl = 3, float c_puct = 3, int device = 0, uint64_t seed = 0,
             int slot_base = 0, int contexts = 0, std::vector<int> devices = {}, std::vector<int> channels = {},
             int precision = AZ_PREC_FP16)
        : G(games), budget(iter_budget), Cs(channels.empty() ? NetConfig::get(2).resblocks() : channels) {
        if (devices.empty()) devices = {device};
        if (contexts <= 0) { const char* e = getenv("AZ_CONTEXTS"); contexts = e ? atoi(e) : 1; }
        if (contexts < 1 || games < 2 * contexts || games % contexts) contexts = 1;
        az_config cfg{};
        cfg.N = Env::N; cfg.K = Env::K; cfg.G = games / contexts;
        cfg.n_Cs = (int)Cs.size();
        for (size_t k = 0; k < Cs.size(); ++k) cfg.Cs[k] = Cs[k];
        cfg.vl = vl; cfg.c_puct = c_puct; cfg.max_budget = iter_budget;
        cfg.eval_mode = AZ_EVAL_TOWER; cfg.noise_mode = AZ_NOISE_DEVICE;
        cfg.seed = seed; cfg.game_slot_base = slot_base; cfg.precision = precision;
        check(az_group_create(&grp, &cfg, devices.data(), (int)devices.size(), contexts));
        n_gpus = (int)devices.size();
    }
    ~SelfPlay() { az_group_destroy(grp); }
    SelfPlay(const SelfPlay&) = delete;
    SelfPlay& operator=(const SelfPlay&) = delete;
    // agent.load() (generate.cpp:73): re-read ckpt_location.txt; on change the new blob goes to the first GPU,
    // RCCL broadcasts it, every GPU folds it on the device, in stream order with the running self-play
    bool load() {
        WeightFile w;
        bool changed = false;
        const std::vector<int>& cs = Cs;
        const std::string path = load_network(w, model_path, changed, [&cs](WeightFile& f) {
            f.N = Env::N; f.c_in = 3; f.c_out = 1; f.Cs = cs;
            f.blob = fresh_blob(Env::N, 3, f.Cs, 0);
        });
        if (changed) {
            if (w.N != Env::N || w.Cs != Cs) throw Error("checkpoint shape does not match the environment and the tower this generator was started with");
            check(az_group_load_weights(grp, w.blob.data(), w.blob.size()));
            model_path = path;
        }
        return changed;
    }
    void begin() { check(az_group_selfplay_begin(grp, budget)); }
    void run(int rounds) { check(az_group_selfplay_run(grp, rounds)); }
    // finished rows of every GPU -> sink(board u8 [3,N,N], policy f32 [1,N,N], reward f32 [2]), GPU-major; waits
    // for everything queued
    int drain(const std::function<void(const ByteTensor&, const FloatTensor&, const FloatTensor&)>& sink) { return drain_impl(sink, false); }
    // The overlapped hand-over: flip() retires the rings that the rounds queued so far append to; drain_retired()
    // waits for those rounds only and hands their rows to the sink while the rounds queued after the flip run.
    void flip() { check(az_group_replay_flip(grp)); }
    int drain_retired(const std::function<void(const ByteTensor&, const FloatTensor&, const FloatTensor&)>& sink) { return drain_impl(sink, true); }
private:
    int drain_impl(const std::function<void(const ByteTensor&, const FloatTensor&, const FloatTensor&)>& sink, bool retired) {
        const int N = Env::N, A = N * N, cap = 4096;
        std::vector<uint8_t> cells((size_t)cap * A), tm(cap);
        std::vector<float> pol((size_t)cap * A), rew((size_t)cap * 2);
        int total = 0;
        for (;;) {
            const int n = retired ? az_group_replay_drain_retired(grp, cap, cells.data(), tm.data(), pol.data(), rew.data())
                                  : az_group_replay_drain(grp, cap, cells.data(), tm.data(), pol.data(), rew.data());
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
public:
    // counters summed over the contexts of every GPU (high-water marks: maximum)
    az_stats stats() { az_stats t{}; check(az_group_get_stats(grp, &t)); return t; }
    az_ctx* context(int i = 0) const { return az_group_context(grp, i); }
    int contexts() const { return az_group_contexts(grp); }
    // `rounds` rounds of every context as one launch of the persistent tower / tree pair, context after context: throws
    // with the library's reason where a context cannot run that form (run() with AZ_PAIR=1 takes lockstep rounds there
    // without saying so).  For a trial at start-up; begin() afterwards puts the games back in step.
    void run_pair(int rounds) { for (int i = 0; i < contexts(); ++i) check(az_selfplay_run_pair(context(i), rounds)); }
    int gpus() const { return n_gpus; }
    const std::string& model() const { return model_path; }
    const int G, budget;
    const std::vector<int> Cs;
private:
    az_group* grp = nullptr;
    int n_gpus = 1;
    std::string model_path;
};

}  // namespace az
