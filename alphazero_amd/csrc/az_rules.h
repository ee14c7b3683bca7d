// az_rules.h -- host-side twin of the device rules in tree.hip (m,n,k-game:
// tictactoe N=3,K=3; gomoku N,K=5).  Used by the C ABI to replay a move
// history into device state and by the C++ Env shim (az/env.hpp).
// Contract from the reference's call sites: mcts.hpp:251-271,427; generate.cpp:75,103.
#pragma once
#include <stdint.h>
#include <string.h>

static inline bool az_rules_line_win(int N, int K, const uint8_t* cells, int action, uint8_t stone) {
    static const int DY[4] = {0, 1, 1, 1};
    static const int DX[4] = {1, 0, 1, -1};
    const int y = action / N, x = action % N;
    for (int d = 0; d < 4; ++d) {
        int run = 1;
        for (int sgn = -1; sgn <= 1; sgn += 2)
            for (int s = 1; s < K; ++s) {
                const int yy = y + sgn * s * DY[d], xx = x + sgn * s * DX[d];
                if (yy < 0 || yy >= N || xx < 0 || xx >= N || cells[yy * N + xx] != stone) break;
                ++run;
            }
        if (run >= K) return true;
    }
    return false;
}

// returns 1 done, 0 running, -1 illegal; reward indexed by absolute player
static inline int az_rules_step(int N, int K, uint8_t* cells, int* to_move, int action, float reward[2]) {
    reward[0] = reward[1] = 0.f;
    if (action < 0 || action >= N * N || cells[action] != 0) return -1;
    const int mover = *to_move;
    cells[action] = (uint8_t)(mover + 1);
    *to_move = 1 - mover;
    if (az_rules_line_win(N, K, cells, action, (uint8_t)(mover + 1))) {
        reward[mover] = 1.f; reward[1 - mover] = -1.f;
        return 1;
    }
    for (int i = 0; i < N * N; ++i) if (cells[i] == 0) return 0;
    return 1;
}

static inline void az_rules_planes(int N, const uint8_t* cells, int to_move, uint8_t* planes) {
    const int A = N * N;
    for (int i = 0; i < A; ++i) {
        planes[i] = cells[i] == 1; planes[A + i] = cells[i] == 2; planes[2 * A + i] = (uint8_t)to_move;
    }
}
