// tensor.hpp -- the POD stand-in for torch::Tensor at the drop-in boundary:
// libtorch is not on the hot path, so the shims exchange owning, contiguous,
// row-major arrays with a shape.
#pragma once
#include <cstdint>
#include <numeric>
#include <stdexcept>
#include <vector>

namespace az {

template <class T>
struct Tensor {
    std::vector<int64_t> shape;
    std::vector<T> v;
    Tensor() = default;
    explicit Tensor(std::vector<int64_t> s, T fill = T())
        : shape(std::move(s)), v((size_t)std::accumulate(shape.begin(), shape.end(), (int64_t)1, std::multiplies<int64_t>()), fill) {}
    int64_t numel() const { return (int64_t)v.size(); }
    const std::vector<int64_t>& sizes() const { return shape; }
    T* data() { return v.data(); }
    const T* data() const { return v.data(); }
    T& operator[](size_t i) { return v[i]; }
    const T& operator[](size_t i) const { return v[i]; }
    bool defined() const { return !shape.empty(); }
};
using ByteTensor = Tensor<uint8_t>;
using FloatTensor = Tensor<float>;

inline int64_t argmax(const FloatTensor& t) {      // torch::argmax: first maximum
    int64_t b = 0;
    for (int64_t i = 1; i < t.numel(); ++i) if (t.v[i] > t.v[b]) b = i;
    return b;
}

struct Error : std::runtime_error { using std::runtime_error::runtime_error; };

}  // namespace az
