// pt2azw.cpp -- the trainer's checkpoint (`torch::save(net, path)`, util.h:108-139) to the flat blob file, in C++
// and without libtorch: what the generator's load_network does with such a file (az/util.hpp: read_pt), as a
// program of its own.   usage: pt2azw <model_N.pt> [out.azw]
#include <cstdio>
#include <string>

#include "../include/az/util.hpp"

int main(int argc, char const* argv[]) {
    if (argc < 2 || argc > 3) { std::fprintf(stderr, "usage: pt2azw <checkpoint.pt> [out.azw]\n"); return 2; }
    try {
        az::WeightFile w;
        az::read_pt(argv[1], w);
        const std::string out = argc == 3 ? argv[2] : std::string(argv[1]) + ".azw";
        az::write_azw(out, w);
        std::printf("wrote %s: N=%d c_in=%d c_out=%d blocks=%zu channels=%d floats=%zu\n", out.c_str(), w.N, w.c_in, w.c_out,
                    w.Cs.size() - 1, w.Cs[0], w.blob.size());
    } catch (const std::exception& e) {
        std::fprintf(stderr, "pt2azw: %s\n", e.what());
        return 1;
    }
    return 0;
}
