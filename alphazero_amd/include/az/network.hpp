// network.hpp -- PVNetwork (network.hpp:103-132) on the HIP tower: same
// constructor signature, forward(x[B,in,N,N]) -> (policy[B,out,N,N] softmax
// probabilities, value[B,2]) (network.hpp:117-123, eval mode).
#pragma once
#include <memory>
#include <tuple>
#include <vector>

#include "../../../include/az_hip.h"
#include "tensor.hpp"

namespace az {

inline void check(int rc) { if (rc < 0) throw Error(az_last_error()); }

class PVNetworkImpl {
public:
    PVNetworkImpl(int board_size, const std::vector<int>& Cs, int in, int out, bool training = false,
                  int K = 3, int P = 1, int max_batch = 256, int device = 0)
        : num_res((int)Cs.size() - 1), N_(board_size) {
        if (in != 3 || out != 1 || K != 3 || P != 1 || training)
            throw Error("PVNetwork: the HIP tower is the eval-mode 3-plane, 1-policy-channel, 3x3 network");
        az_config cfg{};
        cfg.device = device; cfg.N = board_size; cfg.K = board_size < 5 ? board_size : 5; cfg.G = max_batch;
        cfg.n_Cs = (int)Cs.size();
        for (size_t i = 0; i < Cs.size(); ++i) cfg.Cs[i] = Cs[i];
        cfg.vl = 3; cfg.c_puct = 3; cfg.max_budget = 1; cfg.eval_mode = AZ_EVAL_TOWER;
        check(az_ctx_create(&ctx_, &cfg));
    }
    ~PVNetworkImpl() { az_ctx_destroy(ctx_); }
    PVNetworkImpl(const PVNetworkImpl&) = delete;
    size_t weights_size() const { return az_weights_size(ctx_); }
    void load_blob(const float* blob, size_t n) { check(az_load_weights(ctx_, blob, n)); }
    void eval() {}
    std::tuple<FloatTensor, FloatTensor> forward(const ByteTensor& x) {
        const int B = (int)x.shape[0];
        FloatTensor policy({B, 1, N_, N_}), value({B, 2});
        check(az_tower_forward(ctx_, x.data(), B, policy.data(), value.data()));
        return {policy, value};
    }
    int num_res;
private:
    int N_;
    az_ctx* ctx_ = nullptr;
};

// TORCH_MODULE-style holder: PVNetwork net(...); net->forward(x); net(x)
class PVNetwork {
public:
    PVNetwork() = default;
    PVNetwork(std::nullptr_t) {}
    template <class... A> explicit PVNetwork(A&&... a) : p_(std::make_shared<PVNetworkImpl>(std::forward<A>(a)...)) {}
    PVNetworkImpl* operator->() const { return p_.get(); }
    std::tuple<FloatTensor, FloatTensor> operator()(const ByteTensor& x) const { return p_->forward(x); }
    explicit operator bool() const { return (bool)p_; }
private:
    std::shared_ptr<PVNetworkImpl> p_;
};

}  // namespace az
