// netconfig.hpp -- NetConfig (netconfig.hpp:8-68): the five channel tables;
// like the reference, get(n) only accepts n == 2 (netconfig.hpp:13-14).
#pragma once
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace az {

class NetConfig {
public:
    static NetConfig& get(int n) {
        if (n != 2) throw std::runtime_error("Go for option 2.");
        static NetConfig singleton(n);
        return singleton;
    }
    std::vector<int> resblocks() const { return Cs_[id_]; }
    std::string channels_to_string() const {
        std::stringstream st;
        st << "Channels: [ ";
        for (int c : Cs_[id_]) st << c << ' ';
        st << "]\n";
        return st.str();
    }
private:
    explicit NetConfig(int n) : id_(n) {
        Cs_ = {std::vector<int>(20, 32), std::vector<int>(10, 32), std::vector<int>(10, 128),
               {256, 128, 64, 32, 16, 8, 4, 2}, {128, 128, 64, 64, 32, 32, 16, 16}};
    }
    int id_;
    std::vector<std::vector<int>> Cs_;
};

}  // namespace az
