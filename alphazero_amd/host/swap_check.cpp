// swap_check.cpp -- the trainer -> generator weight hand-off (util.h:60-139, generate.cpp:73) against a
// running self-play: ckpt_location.txt is rewritten under flock(LOCK_EX) as save_network does, the
// generator's load() notices, the blob goes through az_group_load_weights (RCCL broadcast + device-side
// fold) and the network that answers changes between two rounds.  Also: a pointer to a file that is neither a blob
// nor a readable libtorch archive (a checkpoint cut short) keeps the current weights instead of ending the process.
//   usage: swap_check <contexts_per_gpu>      (run in an empty directory)
#include <sys/file.h>
#include <unistd.h>

#include <cstdio>
#include <fstream>

#include "../include/az/env.hpp"
#include "../include/az/selfplay.hpp"

using Env = az::Gomoku9;

static void point_at(const std::string& path) {                  // util.h:122-138
    const int fd = open("ckpt_location.txt", O_WRONLY | O_CREAT, 0644);
    flock(fd, LOCK_EX);
    { std::ofstream f("ckpt_location.txt", std::ios::trunc); f << path; }
    flock(fd, LOCK_UN);
    close(fd);
}

static std::vector<float> answer(az_ctx* c, const az::ByteTensor& x) {
    std::vector<float> out(2 * 81 + 4);
    az::check(az_tower_forward(c, x.data(), 2, out.data(), out.data() + 2 * 81));
    return out;
}

#define REQUIRE(cond) do { if (!(cond)) { printf("FAILED: %s (line %d)\n", #cond, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int contexts = argc > 1 ? atoi(argv[1]) : 1;
    const std::vector<int> Cs = az::NetConfig::get(2).resblocks();
    az::WeightFile A, B;
    A.N = B.N = Env::N; A.c_in = B.c_in = 3; A.c_out = B.c_out = 1; A.Cs = B.Cs = Cs;
    A.blob = az::fresh_blob(Env::N, 3, Cs, 1);
    B.blob = az::fresh_blob(Env::N, 3, Cs, 2);
    az::write_azw("model_1.pt.azw", A);
    az::write_azw("model_2.pt.azw", B);
    { std::ofstream f("model_3.pt", std::ios::binary); f << "PK\x03\x04 not a blob"; }      // starts like an archive, is none
    point_at("model_1.pt");

    az::SelfPlay<Env> agent(32, 16, 3, 3, 0, 7, 0, contexts);
    REQUIRE(agent.contexts() == contexts);
    REQUIRE(agent.load());
    REQUIRE(agent.model() == "model_1.pt");
    agent.begin();
    agent.run(16 * 3);
    az::ByteTensor x({2, 3, Env::N, Env::N});
    x.v[40] = 1; x.v[81 + 41] = 1; x.v[243 + 3] = 1; for (int i = 0; i < 81; ++i) x.v[243 + 162 + i] = 1;
    az_ctx* first = agent.context(0);
    az_ctx* last = agent.context(agent.contexts() - 1);
    const std::vector<float> a0 = answer(first, x), a1 = answer(last, x);
    REQUIRE(a0 == a1);
    REQUIRE(!agent.load());                                      // pointer unchanged
    point_at("model_3.pt");                                      // an unreadable checkpoint
    REQUIRE(!agent.load());
    REQUIRE(agent.model() == "model_1.pt");
    agent.run(16);
    REQUIRE(answer(first, x) == a0);
    point_at("model_2.pt");                                      // the next checkpoint, mid-run
    REQUIRE(agent.load());
    REQUIRE(agent.model() == "model_2.pt");
    agent.run(16 * 3);
    const std::vector<float> b0 = answer(first, x), b1 = answer(last, x);
    REQUIRE(b0 == b1);
    REQUIRE(b0 != a0);
    double diff = 0;
    for (size_t i = 0; i < a0.size(); ++i) diff = std::max(diff, (double)std::fabs(a0[i] - b0[i]));
    // the swapped-in network is exactly blob B: same bits as a network that only ever saw B
    az::PVNetwork net(Env::N, Cs, 3, 1);
    net->load_blob(B.blob.data(), B.blob.size());
    az::FloatTensor p, v;
    std::tie(p, v) = net(x);
    for (int i = 0; i < 2 * 81; ++i) REQUIRE(p.v[(size_t)i] == b0[(size_t)i]);
    for (int i = 0; i < 4; ++i) REQUIRE(v.v[(size_t)i] == b0[(size_t)(2 * 81 + i)]);
    const az_stats st = agent.stats();
    REQUIRE(st.moves > 0 && st.error_games == 0);
    int rows = agent.drain([](const az::ByteTensor&, const az::FloatTensor&, const az::FloatTensor&) {});
    printf("swap ok: contexts %d, max |answer A - answer B| = %.4g, moves %lld, rows %d\n", contexts, diff, (long long)st.moves, rows);
    return 0;
}
