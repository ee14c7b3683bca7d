// replay.hpp -- the generator half of ReplayBuffer<Env> (replay.hpp:41-114):
// temporary_append / send_reward emitting, per buffered (state, policy), the 8
// dihedral variants as 3-frame multipart messages
//   [state u8 C*N*N][policy f32 c_out*N*N][reward f32 numel]
// on a ZMQ_PUSH bound to tcp://IP:port.  The trainer half (receive, get_batch,
// load, save) stays the reference's own code.
#pragma once
#include <deque>
#include <iostream>
#include <memory>
#include <string>
#include <tuple>
#include <vector>

#include "tensor.hpp"
#include "zmtp.hpp"

namespace az {

// rot90(r, {1,2}) followed optionally by flip(1) of a [C,N,N] array (replay.hpp:82-99)
template <class T>
inline Tensor<T> sym_variant(const Tensor<T>& x, int r, bool flip) {
    const int C = (int)x.shape[0], N = (int)x.shape[1];
    Tensor<T> out(x.shape);
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < N; ++j) {
                const int ii = flip ? N - 1 - i : i;
                int sy, sx;
                if (r == 0) { sy = ii; sx = j; }
                else if (r == 1) { sy = j; sx = N - 1 - ii; }
                else if (r == 2) { sy = N - 1 - ii; sx = N - 1 - j; }
                else { sy = N - 1 - j; sx = ii; }
                out.v[((size_t)c * N + i) * N + j] = x.v[((size_t)c * N + sy) * N + sx];
            }
    return out;
}

template <class Env>
class ReplayBuffer {
    enum { GEN, TRN };
public:
    ReplayBuffer(int ID, const std::string& IP, const std::string& port, int max = 1, int train_threshold = 0) {
        if (max < train_threshold)
            throw std::runtime_error("Max buffer size should be greater than the training threshold.");
        if (ID != GEN) throw std::runtime_error("the trainer half of ReplayBuffer stays the reference's (replay.hpp:120-279)");
        const std::string path = "tcp://" + IP + ":" + port;
        std::cout << "Path: " << path << std::endl;
        socket_.reset(new ZmtpPush(path));
    }
    void temporary_append(const ByteTensor& state, const FloatTensor& policy) { temp.emplace_back(state, policy); }
    void send_reward(const FloatTensor& reward) {
        while (!temp.empty()) {
            ByteTensor state; FloatTensor policy;
            std::tie(state, policy) = std::move(temp.front());
            temp.pop_front();
            for (int r = 0; r < 4; ++r)
                for (int f = 0; f < 2; ++f) {
                    const ByteTensor s = sym_variant(state, r, f != 0);
                    const FloatTensor p = sym_variant(policy, r, f != 0);
                    socket_->send_multipart({{s.data(), (size_t)s.numel()},
                                             {p.data(), (size_t)p.numel() * sizeof(float)},
                                             {reward.data(), (size_t)reward.numel() * sizeof(float)}});
                }
        }
    }
    int port() const { return socket_->port(); }
    size_t pending() const { return temp.size(); }
private:
    std::deque<std::tuple<ByteTensor, FloatTensor>> temp;
    std::unique_ptr<ZmtpPush> socket_;
};

}  // namespace az
