/*
 * ref_env.hpp -- torch-typed environment adapter so that the reference's
 * UNMODIFIED MCTS<Env> / Evaluator<Env> templates can be instantiated in the
 * oracle harness.  TEST INFRASTRUCTURE ONLY.
 *
 * The reference's own environments are git submodules that are empty in the
 * checkout; this adapter implements the member set those templates call
 * (mcts.hpp:23-25,51,124,251,264,271,387-389,427; evaluator.hpp:30-33,53-55,
 * 61-63; node.hpp:11-14) on top of the oracle's own rules (az_oracle.cpp).
 */
#pragma once
#include <array>
#include <tuple>
#include <vector>
#include <torch/torch.h>
#include "az_oracle.h"

struct MnkState {
    torch::Tensor board;                       // uint8 [3,N,N]
    std::vector<std::array<int, 2>> id;        // move history
    int player = 0;
    std::vector<uint8_t> cells;
};

class MnkEnv {
public:
    using state_type = MnkState;
    using action_type = std::array<int, 2>;
    using id_type = std::vector<action_type>;
    using board_type = torch::Tensor;
    using reward_type = torch::Tensor;
    using shape_type = int64_t;

    static int& cfg_N() { static int n = 3; return n; }
    static int& cfg_K() { static int k = 3; return k; }
    static MnkEnv& get() { static MnkEnv e; return e; }

    int get_board_size() const { return cfg_N(); }
    int get_state_channels() const { return 3; }
    int get_action_channels() const { return 1; }
    std::vector<int64_t> get_board_shape() const { return {3, cfg_N(), cfg_N()}; }
    std::vector<int64_t> get_action_shape() const { return {1, cfg_N(), cfg_N()}; }

    state_type make(const std::vector<uint8_t>& cells, int player, id_type id) const {
        const int N = cfg_N();
        state_type s;
        s.cells = cells; s.player = player; s.id = std::move(id);
        s.board = torch::empty({3, N, N}, torch::kUInt8);
        azo_env_planes(N, cells.data(), player, s.board.data_ptr<uint8_t>());
        return s;
    }
    state_type reset() const {
        std::vector<uint8_t> cells((size_t)cfg_N() * cfg_N(), 0);
        return make(cells, 0, {});
    }
    std::tuple<state_type, torch::Tensor, bool> step(const state_type& s, const action_type& a) const {
        const int N = cfg_N();
        std::vector<uint8_t> cells = s.cells;
        int to_move = s.player;
        float r[2];
        const int done = azo_env_step(N, cfg_K(), cells.data(), &to_move, a[0] * N + a[1], r);
        if (done < 0) throw std::runtime_error("MnkEnv::step illegal action");
        id_type id = s.id;
        id.push_back(a);
        torch::Tensor reward = torch::empty({2});
        reward[0] = r[0]; reward[1] = r[1];
        return {make(cells, to_move, std::move(id)), reward, done == 1};
    }
    int get_player(const state_type& s) const { return s.player; }
    const board_type& get_board(const state_type& s) const { return s.board; }
    const id_type& get_id(const state_type& s) const { return s.id; }
    std::vector<action_type> possible_actions(const state_type& s, int) const {
        const int N = cfg_N();
        std::vector<action_type> out;
        for (int i = 0; i < N * N; ++i)
            if (s.cells[i] == 0) out.push_back({i / N, i % N});
        return out;
    }
};
