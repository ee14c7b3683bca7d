// env.hpp -- the Env concept the reference's MCTS / Evaluator / ReplayBuffer
// templates are written against (call sites: mcts.hpp:23-25,51,124,251,264,271,
// 387-389,427; evaluator.hpp:30-33,53-55,61-63; replay.hpp:122-126;
// generate.cpp:53-54,75,80,102-106), for the m,n,k-game family.  The
// reference's own environments are git submodules that are empty in its tree;
// these rules are this build's (see az_rules.h), with torch::Tensor replaced
// by az::Tensor.
#pragma once
#include <array>
#include <sstream>
#include <tuple>
#include <vector>

#include "../../csrc/az_blokus_rules.h"
#include "../../csrc/az_rules.h"
#include "tensor.hpp"

namespace az {

using Action = std::array<int, 2>;

struct State {
    ByteTensor board;                 // uint8 [3,N,N]
    std::vector<Action> id;           // move history since reset
    int player = 0;
    std::vector<uint8_t> cells;       // 0 empty, 1 player 0, 2 player 1
    bool done = false;
};

template <int N_, int K_>
class MnkEnv {
public:
    using state_type = State;
    using action_type = Action;
    using id_type = std::vector<Action>;
    using board_type = ByteTensor;
    using reward_type = FloatTensor;
    using shape_type = int64_t;
    static constexpr int N = N_, K = K_;

    static MnkEnv& get() { static MnkEnv e; return e; }
    int get_board_size() const { return N; }
    int get_state_channels() const { return 3; }
    int get_action_channels() const { return 1; }
    std::vector<int64_t> get_board_shape() const { return {3, N, N}; }
    std::vector<int64_t> get_action_shape() const { return {1, N, N}; }

    state_type reset() const {
        state_type s;
        s.cells.assign((size_t)N * N, 0);
        refresh(s);
        return s;
    }
    std::tuple<state_type, reward_type, bool> step(const state_type& s, const action_type& a) const {
        state_type n = s;
        reward_type r({2});
        const int d = az_rules_step(N, K, n.cells.data(), &n.player, a[0] * N + a[1], r.data());
        if (d < 0) throw Error("Env::step: illegal action");
        n.id.push_back(a);
        n.done = d == 1;
        refresh(n);
        return {n, r, d == 1};
    }
    int get_player(const state_type& s) const { return s.player; }
    const board_type& get_board(const state_type& s) const { return s.board; }
    const id_type& get_id(const state_type& s) const { return s.id; }
    std::vector<action_type> possible_actions(const state_type& s, int) const {
        std::vector<action_type> out;
        for (int i = 0; i < N * N; ++i) if (s.cells[i] == 0) out.push_back({i / N, i % N});
        return out;
    }
    std::stringstream to_string(const state_type& s) const { return render(s, {-1, -1}); }
    std::stringstream to_string(const state_type& s, const action_type& last) const { return render(s, last); }

private:
    static void refresh(state_type& s) {
        s.board = ByteTensor({3, N, N});
        az_rules_planes(N, s.cells.data(), s.player, s.board.data());
    }
    static std::stringstream render(const state_type& s, const action_type& last) {
        std::stringstream ss;
        for (int y = 0; y < N; ++y) {
            for (int x = 0; x < N; ++x) {
                const uint8_t c = s.cells[y * N + x];
                const bool hl = y == last[0] && x == last[1];
                ss << (c == 1 ? (hl ? '0' : 'O') : c == 2 ? (hl ? '%' : 'X') : '.') << ' ';
            }
            ss << '\n';
        }
        return ss;
    }
};

// Blokus behind the same Env concept (call sites as above), on the library's own statement of the rules
// (az_blokus_rules.h; the reference's Blokus submodule is an empty directory): actions are 3-tuples
// {orientation, y, x} -- the fossil the reference keeps in Node::operator<< (node.hpp:108-110) -- the action
// shape is [91, 20, 20], the board [5, 20, 20], the reward has four entries.
using Action3 = std::array<int, 3>;

struct BlokusState {
    ByteTensor board;                 // uint8 [5,20,20]
    std::vector<Action3> id;          // move history since reset
    int player = 0;
    AzbGame game;
    bool done = false;
};

class Blokus {
public:
    using state_type = BlokusState;
    using action_type = Action3;
    using id_type = std::vector<Action3>;
    using board_type = ByteTensor;
    using reward_type = FloatTensor;
    using shape_type = int64_t;
    static constexpr int N = AZB_N;

    static Blokus& get() { static Blokus e; return e; }
    int get_board_size() const { return N; }
    int get_state_channels() const { return AZB_PLANES; }
    int get_action_channels() const { return AZB_ORIENTS; }
    std::vector<int64_t> get_board_shape() const { return {AZB_PLANES, N, N}; }
    std::vector<int64_t> get_action_shape() const { return {AZB_ORIENTS, N, N}; }

    state_type reset() const {
        state_type s;
        azb_reset(&s.game);
        refresh(s);
        return s;
    }
    std::tuple<state_type, reward_type, bool> step(const state_type& s, const action_type& a) const {
        state_type n = s;
        reward_type r({AZB_PLAYERS});
        const int d = azb_step(&n.game, table(), (a[0] * N + a[1]) * N + a[2], r.data());
        if (d < 0) throw Error("Env::step: illegal action");
        n.id.push_back(a);
        n.done = d == 1;
        refresh(n);
        return {n, r, d == 1};
    }
    int get_player(const state_type& s) const { return s.player; }
    const board_type& get_board(const state_type& s) const { return s.board; }
    const id_type& get_id(const state_type& s) const { return s.id; }
    std::vector<action_type> possible_actions(const state_type& s, int player) const {
        std::vector<action_type> out;
        if (s.done) return out;
        for (int a = 0; a < AZB_ACTIONS; ++a)
            if (azb_legal(&s.game, table(), player, a)) out.push_back({a / AZB_CELLS, a % AZB_CELLS / N, a % N});
        return out;
    }
    std::stringstream to_string(const state_type& s) const {
        std::stringstream ss;
        for (int y = 0; y < N; ++y) {
            for (int x = 0; x < N; ++x) ss << ".ABCD"[s.game.cells[y * N + x]] << ' ';
            ss << '\n';
        }
        return ss;
    }
    std::stringstream to_string(const state_type& s, const action_type&) const { return to_string(s); }

private:
    static const AzbOrient* table() {
        static AzbOrient T[AZB_ORIENTS + 8];
        static const int n = azb_build_orients(T);
        (void)n;
        return T;
    }
    static void refresh(state_type& s) {
        s.player = s.game.to_move;
        s.board = ByteTensor({AZB_PLANES, N, N});
        azb_planes(&s.game, s.board.data());
    }
};

using TicTacToe = MnkEnv<3, 3>;
using Gomoku9 = MnkEnv<9, 5>;
using Gomoku = MnkEnv<15, 5>;

}  // namespace az
