// zmtp_send.cpp -- sends one buffered game through ReplayBuffer(GEN) so that a
// real libzmq PULL socket can check the wire format (tests/test_wire.py).
// usage: zmtp_send <port> <N> <n_rows>; rows are a deterministic pattern.
#include <cstdio>
#include <cstdlib>

#include "../include/az/env.hpp"
#include "../include/az/replay.hpp"

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int N = atoi(argv[2]), rows = atoi(argv[3]);
    az::ReplayBuffer<az::TicTacToe> buffer(0, "127.0.0.1", argv[1]);
    printf("port %d\n", buffer.port());
    fflush(stdout);
    for (int r = 0; r < rows; ++r) {
        az::ByteTensor s({3, N, N});
        az::FloatTensor p({1, N, N});
        for (int i = 0; i < 3 * N * N; ++i) s.v[(size_t)i] = (uint8_t)((i * 7 + r) % 3 == 0);
        for (int i = 0; i < N * N; ++i) p.v[(size_t)i] = (float)(i + 1 + 100 * r) / 1000.f;
        buffer.temporary_append(s, p);
    }
    az::FloatTensor reward({2});
    reward.v[0] = 1.f; reward.v[1] = -1.f;
    buffer.send_reward(reward);
    printf("sent\n");
    fflush(stdout);
    // keep the socket open until the reader closes stdin
    getchar();
    return 0;
}
