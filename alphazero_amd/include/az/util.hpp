// util.hpp -- the checkpoint pointer protocol of util.h:60-139 for the flat
// weight blob the HIP tower consumes.  `ckpt_location.txt` (read under
// flock(LOCK_SH), util.h:63-87) names the newest `<prefix>_<ver>.pt`; the blob
// is looked up as `<that path>.azw` (a converted twin, tools/pt2blob.py), or the
// named file is read directly: as a blob, or as the trainer's own libtorch
// archive (`torch::save(net, path)`, util.h:108-139; ptarchive.hpp, no libtorch).
#pragma once
#include <sys/file.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "ptarchive.hpp"
#include "tensor.hpp"

namespace az {

struct WeightFile {
    int N = 0, c_in = 0, c_out = 0;
    std::vector<int> Cs;
    std::vector<float> blob;
};

inline bool read_azw(const std::string& path, WeightFile& w) {
    std::ifstream f(path, std::ios::binary);
    char magic[4];
    if (!f.read(magic, 4) || memcmp(magic, "AZW1", 4) != 0) return false;
    uint32_t h[4];
    if (!f.read((char*)h, 16)) return false;
    w.N = (int)h[0]; w.c_in = (int)h[1]; w.c_out = (int)h[2];
    w.Cs.resize(h[3]);
    for (auto& c : w.Cs) { uint32_t x; if (!f.read((char*)&x, 4)) return false; c = (int)x; }
    uint64_t n;
    if (!f.read((char*)&n, 8)) return false;
    w.blob.resize(n);
    return (bool)f.read((char*)w.blob.data(), (std::streamsize)(n * 4));
}

inline void write_azw(const std::string& path, const WeightFile& w) {
    std::ofstream f(path, std::ios::binary);
    f.write("AZW1", 4);
    const uint32_t h[4] = {(uint32_t)w.N, (uint32_t)w.c_in, (uint32_t)w.c_out, (uint32_t)w.Cs.size()};
    f.write((const char*)h, 16);
    for (int c : w.Cs) { const uint32_t x = (uint32_t)c; f.write((const char*)&x, 4); }
    const uint64_t n = w.blob.size();
    f.write((const char*)&n, 8);
    f.write((const char*)w.blob.data(), (std::streamsize)(n * 4));
}

// The trainer's checkpoint (torch::save(net, path), util.h:125) -> blob: tensors by their registered names
// (network.hpp:106-114: CBlock, PHead, VHead, RBlock_i; running_mean / running_var are buffers of the BatchNorms),
// in the blob's order; the tower's shape is read off the tensors.  Throws pt::Error if the file is not such an archive.
inline void read_pt(const std::string& path, WeightFile& w) {
    pt::Archive a(path);
    auto shape = [&](const std::string& n) -> const std::vector<int64_t>& { return a.at(n).shape; };
    if (shape("CBlock.conv.weight").size() != 4 || shape("VHead.fc1.weight").size() != 2)
        throw pt::Error(path + ": CBlock.conv.weight / VHead.fc1.weight do not have a convolution's / a linear layer's shape");
    w.Cs.clear();
    w.Cs.push_back((int)shape("CBlock.conv.weight")[0]);
    w.c_in = (int)shape("CBlock.conv.weight")[1];
    for (int i = 0; a.has("RBlock_" + std::to_string(i) + ".conv1.weight"); ++i)
        w.Cs.push_back((int)shape("RBlock_" + std::to_string(i) + ".conv1.weight")[0]);
    w.c_out = (int)shape("PHead.conv.weight")[0];
    const int A = (int)shape("VHead.fc1.weight")[1];
    w.N = (int)std::lround(std::sqrt((double)A));
    if (w.N * w.N != A || w.Cs.size() < 2) throw pt::Error(path + ": not a PVNetwork checkpoint (fc1 width / ResBlocks)");
    w.blob.clear();
    auto put = [&](const std::string& n, std::vector<int64_t> want) {
        const pt::Tensor& t = a.at(n);
        if (t.shape != want) throw pt::Error(path + ": tensor " + n + " has an unexpected shape");
        w.blob.insert(w.blob.end(), t.v.begin(), t.v.end());
    };
    auto conv = [&](const std::string& m, int64_t ci, int64_t co, int64_t k) { put(m + ".weight", {co, ci, k, k}); put(m + ".bias", {co}); };
    auto bn = [&](const std::string& m, int64_t ch) {
        put(m + ".weight", {ch}); put(m + ".bias", {ch}); put(m + ".running_mean", {ch}); put(m + ".running_var", {ch});
    };
    const size_t num_res = w.Cs.size() - 1;
    const int64_t Ch = w.Cs[num_res - 1];
    conv("CBlock.conv", w.c_in, w.Cs[0], 3); bn("CBlock.bn", w.Cs[0]);
    conv("PHead.conv", Ch, w.c_out, 1);
    conv("VHead.conv", Ch, 1, 1); bn("VHead.bn", 1);
    const int64_t vd = shape("VHead.fc2.weight").empty() ? 2 : shape("VHead.fc2.weight")[0];
    put("VHead.fc1.weight", {64, A}); put("VHead.fc1.bias", {64});
    put("VHead.fc2.weight", {vd, 64}); put("VHead.fc2.bias", {vd});
    for (size_t i = 0; i < num_res; ++i) {
        const std::string r = "RBlock_" + std::to_string(i);
        conv(r + ".conv1", w.Cs[i], w.Cs[i + 1], 3); conv(r + ".conv2", w.Cs[i + 1], w.Cs[i + 1], 3);
        bn(r + ".bn1", w.Cs[i + 1]); bn(r + ".bn2", w.Cs[i + 1]);
    }
}

// A freshly constructed network in blob order (the reference saves one when the named file is missing,
// util.h:94-97): module registration order CBlock, PHead, VHead, RBlock_i (network.hpp:106-114) with torch's
// default initialisation -- convolutions and linear layers U(+-1/sqrt(fan_in)) for weight and bias, BatchNorm
// weight 1, bias 0, running_mean 0, running_var 1.
inline std::vector<float> fresh_blob(int N, int c_in, const std::vector<int>& Cs, uint32_t seed) {
    std::mt19937 g(seed);
    std::vector<float> b;
    auto uniform = [&](size_t n, size_t fan_in) {
        const float k = 1.0f / std::sqrt((float)fan_in);
        std::uniform_real_distribution<float> u(-k, k);
        for (size_t i = 0; i < n; ++i) b.push_back(u(g));
    };
    auto conv = [&](size_t ci, size_t co, size_t ks) { uniform(co * ci * ks * ks, ci * ks * ks); uniform(co, ci * ks * ks); };
    auto linear = [&](size_t in, size_t out) { uniform(out * in, in); uniform(out, in); };
    auto bn = [&](size_t ch) {
        b.insert(b.end(), ch, 1.0f); b.insert(b.end(), ch, 0.0f);       // weight, bias
        b.insert(b.end(), ch, 0.0f); b.insert(b.end(), ch, 1.0f);       // running_mean, running_var
    };
    const size_t num_res = Cs.size() - 1, Ch = (size_t)Cs[num_res - 1], A = (size_t)N * N;
    conv((size_t)c_in, (size_t)Cs[0], 3); bn((size_t)Cs[0]);
    conv(Ch, 1, 1);                                                   // PHead
    conv(Ch, 1, 1); bn(1); linear(A, 64); linear(64, 2);              // VHead
    for (size_t i = 0; i < num_res; ++i) {
        conv((size_t)Cs[i], (size_t)Cs[i + 1], 3); conv((size_t)Cs[i + 1], (size_t)Cs[i + 1], 3);
        bn((size_t)Cs[i + 1]); bn((size_t)Cs[i + 1]);
    }
    return b;
}

// util.h:60-105.  Returns the path named by ckpt_location.txt; fills `w` when
// the path differs from `current`.  `fresh` supplies weights if nothing exists.
template <class MakeFresh>
inline std::string load_network(WeightFile& w, const std::string& current, bool& changed, MakeFresh fresh) {
    std::cout << "Loading network." << std::endl;
    const int fd = open("ckpt_location.txt", O_RDONLY);
    if (fd >= 0) flock(fd, LOCK_SH);
    std::ifstream loc("ckpt_location.txt");
    if (!loc.is_open()) {
        if (fd >= 0) { flock(fd, LOCK_UN); close(fd); }
        char* cwd = getcwd(nullptr, 0);
        std::stringstream ss;
        ss << "\n* ERROR *\n\"ckpt_location.txt\" must be in the working directory.\n"
           << "Furthermore, the path that \"ckpt_location.txt\" points to must exist.\n"
           << "Current working directory: " << (cwd ? cwd : "?") << "\n";
        free(cwd);
        throw Error(ss.str());
    }
    std::string export_path;
    std::getline(loc, export_path);
    loc.close();
    if (fd >= 0) { flock(fd, LOCK_UN); close(fd); }
    changed = false;
    if (export_path == current) {
        std::cout << "No change in network. Resuming with network " << current << std::endl;
        return current;
    }
    if (read_azw(export_path + ".azw", w) || read_azw(export_path, w)) {
        std::cout << "Network loaded.\nPath: " << export_path << std::endl;
    } else if (std::ifstream(export_path).good()) {
        // the trainer's own file: a libtorch archive (util.h:108-139), read as it is
        try {
            read_pt(export_path, w);
            std::cout << "Network loaded.\nPath: " << export_path << std::endl;
        } catch (const pt::Error& e) {
            // not (yet) a readable archive: a generator with weights keeps playing with them and looks again at its
            // next load(); one without has nothing to play with
            if (current.empty()) throw Error("checkpoint " + export_path + " cannot be read: " + e.what());
            std::cout << "Checkpoint " << export_path << " cannot be read (" << e.what() << "). Resuming with network " << current << std::endl;
            return current;
        }
    } else {
        std::cout << "Model path provided by 'ckpt_location.txt' doesn't exist. Creating a new one." << std::endl;
        fresh(w);
        write_azw(export_path + ".azw", w);
    }
    changed = true;
    const auto now = std::chrono::system_clock::to_time_t(std::chrono::system_clock::now());
    std::cout << "Current time: " << ctime(&now);
    return export_path;
}

// util.h:42-58
inline std::vector<float> dirichlet(int size, float alpha = 0.05f) {
    static std::mt19937 rng = std::mt19937(std::random_device{}());
    std::gamma_distribution<float> dist(alpha);
    std::vector<float> vec((size_t)size);
    for (auto& f : vec) f = dist(rng);
    double sum = 0.;
    for (float f : vec) sum += f;
    for (float& f : vec) f = (float)(f / sum);
    return vec;
}

// util.h:141-167
inline std::stringstream visualize_stream(const FloatTensor& policy) {
    static const std::string alphabet = "ABCDEFGHIJKLMNOPQRSTUVWXYZ";
    std::stringstream st;
    const int size = (int)policy.shape[1];
    st << "   ";
    for (char c : alphabet.substr(0, (size_t)size)) st << c << ' ';
    st << '\n';
    for (int i = 0; i < size; ++i) {
        st << (i < 10 ? " " : "") << i << ' ';
        for (int j = 0; j < size; ++j) {
            const int lvl = (int)(policy.v[(size_t)i * size + j] * 10);
            if (lvl != 0) st << lvl << ' '; else st << ". ";
        }
        st << '\n';
    }
    return st;
}

}  // namespace az
