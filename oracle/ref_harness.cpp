/*
 * ref_harness.cpp -- drives the reference's UNMODIFIED headers
 * (network.hpp, netconfig.hpp, node.hpp, evaluator.hpp, mcts.hpp, util.h,
 * included from the read-only reference checkout, never copied) to
 *   tower : evaluate boards with PVNetwork::forward   (network.hpp:117-123)
 *   mcts  : run MCTS<Env>::search_probs at nthreads=1 (mcts.hpp:105-155) and
 *           dump trees, output policies and the (board -> policy,value) table
 *   bench : time the generate.cpp:72-120 loop around the real MCTS on CPU
 * TEST INFRASTRUCTURE ONLY.  Built into oracle/_ref/az_ref by oracle/Makefile.
 *
 * The reference targets 2019 libtorch where torch::nn::BatchNorm existed
 * (network.hpp:13); the alias below is the only accommodation.
 */
#include <torch/torch.h>
namespace torch { namespace nn { using BatchNorm = BatchNorm2d; } }

#include <queue>
#include "mcts.hpp"          // pulls evaluator.hpp, node.hpp, util.h, network.hpp, netconfig.hpp

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <random>
#include <string>
#include <unistd.h>

#include "ref_env.hpp"

using Env = MnkEnv;

/* ------------------------------------------------------------ npy writer -- */
template <class T> struct npy_descr;
template <> struct npy_descr<float> { static const char* s() { return "<f4"; } };
template <> struct npy_descr<int32_t> { static const char* s() { return "<i4"; } };
template <> struct npy_descr<uint8_t> { static const char* s() { return "|u1"; } };
template <class T>
static void save_npy(const std::string& path, const std::vector<T>& data, std::vector<size_t> shape) {
    std::string sh = "(";
    for (size_t i = 0; i < shape.size(); ++i) sh += std::to_string(shape[i]) + ",";
    sh += ")";
    std::string hdr = std::string("{'descr': '") + npy_descr<T>::s() + "', 'fortran_order': False, 'shape': " + sh + ", }";
    while ((10 + hdr.size() + 1) % 64) hdr += ' ';
    hdr += '\n';
    std::ofstream f(path, std::ios::binary);
    const char magic[] = "\x93NUMPY\x01\x00";
    f.write(magic, 8);
    uint16_t hl = (uint16_t)hdr.size();
    f.write((const char*)&hl, 2);
    f.write(hdr.data(), hdr.size());
    f.write((const char*)data.data(), data.size() * sizeof(T));
}
template <class T>
static std::vector<T> load_raw(const std::string& path, size_t n) {
    std::vector<T> v(n);
    std::ifstream f(path, std::ios::binary);
    f.read((char*)v.data(), n * sizeof(T));
    if ((size_t)f.gcount() != n * sizeof(T)) { fprintf(stderr, "short read %s\n", path.c_str()); exit(2); }
    return v;
}

/* ------------------------------------------------- blob -> PVNetwork params */
static void copy_in(torch::Tensor t, const float*& p) {
    torch::NoGradGuard g;
    auto src = torch::from_blob((void*)p, t.sizes(), torch::kFloat32);
    t.copy_(src);
    p += t.numel();
}
static void fill_from_blob(PVNetwork& net, const std::vector<float>& blob) {
    std::map<std::string, torch::Tensor> m;
    for (auto& kv : net->named_parameters()) m[kv.key()] = kv.value();
    for (auto& kv : net->named_buffers()) m[kv.key()] = kv.value();
    const float* p = blob.data();
    auto conv = [&](const std::string& n) { copy_in(m.at(n + ".weight"), p); copy_in(m.at(n + ".bias"), p); };
    auto bn = [&](const std::string& n) {
        copy_in(m.at(n + ".weight"), p); copy_in(m.at(n + ".bias"), p);
        copy_in(m.at(n + ".running_mean"), p); copy_in(m.at(n + ".running_var"), p);
    };
    conv("CBlock.conv"); bn("CBlock.bn");
    conv("PHead.conv");
    conv("VHead.conv"); bn("VHead.bn");
    copy_in(m.at("VHead.fc1.weight"), p); copy_in(m.at("VHead.fc1.bias"), p);
    copy_in(m.at("VHead.fc2.weight"), p); copy_in(m.at("VHead.fc2.bias"), p);
    for (int i = 0; i < net->num_res; ++i) {
        const std::string r = "RBlock_" + std::to_string(i);
        conv(r + ".conv1"); conv(r + ".conv2"); bn(r + ".bn1"); bn(r + ".bn2");
    }
    if ((size_t)(p - blob.data()) != blob.size()) { fprintf(stderr, "blob size mismatch\n"); exit(2); }
}
static void dump_to_blob(PVNetwork& net, std::vector<float>& blob) {
    std::map<std::string, torch::Tensor> m;
    for (auto& kv : net->named_parameters()) m[kv.key()] = kv.value();
    for (auto& kv : net->named_buffers()) m[kv.key()] = kv.value();
    auto put = [&](const std::string& n) {
        auto t = m.at(n).contiguous().to(torch::kFloat32);
        blob.insert(blob.end(), t.data_ptr<float>(), t.data_ptr<float>() + t.numel());
    };
    auto conv = [&](const std::string& n) { put(n + ".weight"); put(n + ".bias"); };
    auto bn = [&](const std::string& n) { put(n + ".weight"); put(n + ".bias"); put(n + ".running_mean"); put(n + ".running_var"); };
    conv("CBlock.conv"); bn("CBlock.bn"); conv("PHead.conv"); conv("VHead.conv"); bn("VHead.bn");
    put("VHead.fc1.weight"); put("VHead.fc1.bias"); put("VHead.fc2.weight"); put("VHead.fc2.bias");
    for (int i = 0; i < net->num_res; ++i) {
        const std::string r = "RBlock_" + std::to_string(i);
        conv(r + ".conv1"); conv(r + ".conv2"); bn(r + ".bn1"); bn(r + ".bn2");
    }
}

/* ------------------------------------------------------------------ tower -- */
static int cmd_tower(int argc, char** argv) {
    if (argc < 9) { fprintf(stderr, "tower N c_in blob boards B out C0 C1 ...\n"); return 2; }
    const int N = atoi(argv[2]), c_in = atoi(argv[3]);
    const std::string blob_path = argv[4], boards_path = argv[5];
    const int B = atoi(argv[6]);
    const std::string out_prefix = argv[7];
    std::vector<int> Cs;
    for (int i = 8; i < argc; ++i) Cs.push_back(atoi(argv[i]));
    const char* co_env = getenv("AZ_REF_COUT");                 /* policy channels (network.hpp:104 `out`), default 1 */
    const int c_out = co_env ? atoi(co_env) : 1;
    PVNetwork net(N, Cs, c_in, c_out);
    size_t n = 0;
    for (auto& p : net->parameters()) n += p.numel();
    for (auto& kv : net->named_buffers())
        if (kv.key().find("num_batches") == std::string::npos) n += kv.value().numel();
    auto blob = load_raw<float>(blob_path, n);
    fill_from_blob(net, blob);
    net->eval();
    torch::NoGradGuard g;
    auto boards = load_raw<uint8_t>(boards_path, (size_t)B * c_in * N * N);
    auto X = torch::from_blob(boards.data(), {B, c_in, N, N}, torch::kUInt8).to(torch::kFloat32);
    std::vector<float> pol((size_t)B * c_out * N * N), val((size_t)B * 2);
    for (int b0 = 0; b0 < B; b0 += 16) {      /* evaluator.hpp:113-114, batch <= 16 */
        const int b1 = std::min(B, b0 + 16);
        torch::Tensor p, v;
        std::tie(p, v) = net(X.slice(0, b0, b1));
        p = p.contiguous(); v = v.contiguous();
        memcpy(pol.data() + (size_t)b0 * c_out * N * N, p.data_ptr<float>(), sizeof(float) * p.numel());
        memcpy(val.data() + (size_t)b0 * 2, v.data_ptr<float>(), sizeof(float) * v.numel());
    }
    save_npy(out_prefix + "_policy.npy", pol, {(size_t)B, (size_t)c_out * N * N});
    save_npy(out_prefix + "_value.npy", val, {(size_t)B, 2});
    return 0;
}

/* ------------------------------------------------------------------- mcts -- */
static void randomise_net(PVNetwork& net, float phead_gain, float fc2_gain) {
    torch::NoGradGuard g;
    for (auto& kv : net->named_buffers()) {
        if (kv.key().find("running_mean") != std::string::npos) kv.value().normal_(0, 0.1);
        if (kv.key().find("running_var") != std::string::npos) kv.value().uniform_(0.5, 1.5);
    }
    for (auto& kv : net->named_parameters()) {
        if (kv.key().find("bn") != std::string::npos && kv.key().find("weight") != std::string::npos) kv.value().uniform_(0.8, 1.2);
        if (kv.key().find("bn") != std::string::npos && kv.key().find("bias") != std::string::npos) kv.value().normal_(0, 0.1);
        if (kv.key() == "PHead.conv.weight") kv.value().mul_(phead_gain);
        if (kv.key() == "VHead.fc2.weight") kv.value().mul_(fc2_gain);
        if (kv.key() == "VHead.bn.bias") kv.value().fill_(0.5);
    }
}

struct TreeDump { std::vector<int32_t> parent, action, n, kind; std::vector<float> v, p; };
static void dump_tree(const std::shared_ptr<Node<Env>>& root, int N, TreeDump& d) {
    std::vector<std::shared_ptr<Node<Env>>> q{root};
    for (size_t qi = 0; qi < q.size(); ++qi) {
        for (auto& kv : q[qi]->children) {
            auto& c = kv.second;
            int kd = 0;
            if (c->terminal) {
                const float r0 = c->reward[0].item<float>(), r1 = c->reward[1].item<float>();
                kd = r0 > 0 ? 2 : (r1 > 0 ? 3 : 4);
            } else if (c->children.size() > 0) kd = 1;
            d.parent.push_back((int32_t)qi); d.action.push_back(kv.first[0] * N + kv.first[1]);
            d.n.push_back(c->n); d.v.push_back(c->v); d.p.push_back(c->p); d.kind.push_back(kd);
            if (kd == 1) q.push_back(c);
        }
    }
}

static uint32_t magic_seed() {
    /* a seed whose first mt19937 output is tiny, so that libstdc++'s
     * uniform_int_distribution<int>(0,k-1) (Lemire: (u64(out)*k)>>32) returns 0
     * for every k <= 512: a tie in MCTS::__choose (mcts.hpp:379-380) then
     * resolves to the first child in map order. */
    for (uint32_t s = 1;; ++s) {
        std::mt19937 g(s);
        const uint32_t out = g();
        if (out != 0 && out < (1u << 22)) return s;
    }
}

static int cmd_mcts(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "mcts N K budget n_moves noise_seed outdir [wseed]\n"); return 2; }
    const int N = atoi(argv[2]), K = atoi(argv[3]), budget = atoi(argv[4]), n_moves = atoi(argv[5]);
    const int noise_seed = atoi(argv[6]);
    const std::string outdir = argv[7];
    const int wseed = argc > 8 ? atoi(argv[8]) : 0;
    Env::cfg_N() = N; Env::cfg_K() = K;
    const int A = N * N;

    /* weights: seeded random init + randomised BN + sharpened heads, saved where
     * the reference Evaluator will load it from (util.h:60-105). */
    if (chdir(outdir.c_str()) != 0) { perror("chdir"); return 2; }
    torch::manual_seed(wseed);
    PVNetwork net(N, NetConfig::get(2).resblocks(), 3, 1);
    randomise_net(net, 30.f, 10.f);
    torch::save(net, "model_0.pt");
    { std::ofstream f("ckpt_location.txt"); f << "model_0.pt"; }
    net->eval();
    torch::NoGradGuard g;
    MCTS<Env> agent(1, 1, 3, 3);         /* nthreads=1; vl=3, c_puct=3 as generate.cpp:51 */
    Env& env = Env::get();
    auto state = env.reset();

    auto run_net = [&](const torch::Tensor& board, std::vector<float>& pol, float* val) {
        auto X = board.to(torch::kFloat32).unsqueeze(0);
        torch::Tensor p, v;
        std::tie(p, v) = net(X);
        p = p.contiguous(); v = v.contiguous();
        pol.assign(p.data_ptr<float>(), p.data_ptr<float>() + A);
        val[0] = v.data_ptr<float>()[0]; val[1] = v.data_ptr<float>()[1];
    };

    /* explicit Dirichlet noise (the reference draws it from random_device) */
    std::vector<float> noise(A);
    { uint64_t st = (uint64_t)noise_seed; azo_dirichlet(&st, A, 0.05f, noise.data()); }

    /* pre-built fresh root == what __make_root builds on a miss (mcts.hpp:392-414),
     * with the noise made an explicit input.  find() then hits target==root and
     * __make_root neither rebuilds it nor calls dirichlet() (mcts.hpp:417). */
    {
        std::vector<float> pol; float val[2];
        run_net(env.get_board(state), pol, val);
        auto root = std::make_shared<Node<Env>>(env.get_id(state), 1, nullptr);
        root->add(state, env.get_player(state), false);
        root->n++;
        int k = 0;
        for (auto& a : env.possible_actions(state, 0)) {
            float prior = pol[a[0] * N + a[1]];
            prior = 0.75 * prior + 0.25 * noise[k++];
            auto id = root->id; id.push_back(a);
            root->children.emplace(a, std::make_shared<Node<Env>>(id, prior, root));
        }
        agent.root = root;
    }

    const uint32_t MAGIC = magic_seed();
    std::map<std::string, std::pair<std::vector<float>, std::array<float, 2>>> table;
    std::vector<int32_t> moves;
    std::vector<float> policies;
    std::vector<int32_t> root_ns, tie_draws;
    int step = 0;
    bool done = false;
    for (; step < n_moves && !done; ++step) {
        agent.rng.seed(MAGIC);
        auto policy = agent.search_probs(state, budget, 0);
        /* how many tie-break draws did the search consume? */
        std::mt19937 probe(MAGIC);
        int draws = -1;
        for (int d = 0; d < 8; ++d) { if (probe == agent.rng) { draws = d; break; } probe(); }
        if (draws < 0 || draws > 1) { fprintf(stderr, "step %d: %d tie draws, fixture invalid\n", step, draws); return 3; }
        tie_draws.push_back(draws);
        auto pc = policy.contiguous();
        policies.insert(policies.end(), pc.data_ptr<float>(), pc.data_ptr<float>() + A);
        root_ns.push_back(agent.root->n);

        /* record every evaluation reachable from this root, self-checking that the
         * children's priors are exactly the recomputed policy */
        std::vector<std::shared_ptr<Node<Env>>> q{agent.root};
        for (size_t qi = 0; qi < q.size(); ++qi) {
            auto nd = q[qi];   /* copy: q grows below */
            const auto& board = nd->state.board;
            std::string key((const char*)board.data_ptr<uint8_t>(), (size_t)3 * A);
            if (!table.count(key)) {
                std::vector<float> pol; std::array<float, 2> val;
                run_net(board, pol, val.data());
                table[key] = {pol, val};
            }
            const auto& pol = table[key].first;
            for (auto& kv : nd->children) {
                if (nd->parent.lock() && kv.second->p != pol[kv.first[0] * N + kv.first[1]]) {
                    fprintf(stderr, "prior mismatch: recomputed policy differs from the evaluator's\n");
                    return 4;
                }
                if (!kv.second->terminal && kv.second->children.size() > 0) q.push_back(kv.second);
            }
        }
        TreeDump d;
        dump_tree(agent.root, N, d);
        const std::string pre = "step" + std::to_string(step) + "_";
        save_npy(pre + "parent.npy", d.parent, {d.parent.size()});
        save_npy(pre + "action.npy", d.action, {d.action.size()});
        save_npy(pre + "n.npy", d.n, {d.n.size()});
        save_npy(pre + "kind.npy", d.kind, {d.kind.size()});
        save_npy(pre + "v.npy", d.v, {d.v.size()});
        save_npy(pre + "p.npy", d.p, {d.p.size()});

        /* move = first arg-max of the policy (generate.cpp:95) */
        int point = 0;
        for (int i = 1; i < A; ++i) if (pc.data_ptr<float>()[i] > pc.data_ptr<float>()[point]) point = i;
        moves.push_back(point);
        torch::Tensor reward;
        std::tie(state, reward, done) = env.step(state, {point / N, point % N});
    }
    std::vector<uint8_t> tp; std::vector<float> tpol, tval;
    for (auto& kv : table) {
        tp.insert(tp.end(), kv.first.begin(), kv.first.end());
        tpol.insert(tpol.end(), kv.second.first.begin(), kv.second.first.end());
        tval.push_back(kv.second.second[0]); tval.push_back(kv.second.second[1]);
    }
    save_npy("table_planes.npy", tp, {table.size(), (size_t)3 * A});
    save_npy("table_policy.npy", tpol, {table.size(), (size_t)A});
    save_npy("table_value.npy", tval, {table.size(), 2});
    save_npy("noise.npy", noise, {(size_t)A});
    save_npy("moves.npy", moves, {moves.size()});
    save_npy("policies.npy", policies, {moves.size(), (size_t)A});
    save_npy("root_n.npy", root_ns, {root_ns.size()});
    save_npy("tie_draws.npy", tie_draws, {tie_draws.size()});
    std::vector<int32_t> meta{N, K, budget, step, (int32_t)done};
    save_npy("meta.npy", meta, {meta.size()});
    std::vector<float> blob;
    dump_to_blob(net, blob);
    { std::ofstream f("weights.bin", std::ios::binary); f.write((const char*)blob.data(), blob.size() * 4); }
    printf("ok steps=%d table=%zu done=%d\n", step, table.size(), (int)done);
    fflush(stdout);
    _exit(0);   /* the reference's singletons join threads in an order that can hang at exit */
}

/* ------------------------------------------------------------------ bench -- */
static int cmd_bench(int argc, char** argv) {
    if (argc < 8) { fprintf(stderr, "bench N K nthreads budget seconds workdir\n"); return 2; }
    const int N = atoi(argv[2]), K = atoi(argv[3]), nthreads = atoi(argv[4]), budget = atoi(argv[5]);
    const double seconds = atof(argv[6]);
    if (chdir(argv[7]) != 0) { perror("chdir"); return 2; }
    Env::cfg_N() = N; Env::cfg_K() = K;
    { std::ofstream f("ckpt_location.txt"); f << "bench_0.pt"; }
    torch::manual_seed(0);
    const int batch_size = 16, vl = 3, c_puct = 3;      /* generate.cpp:51 */
    MCTS<Env> agent(nthreads, batch_size, vl, c_puct);
    Env& env = Env::get();
    std::mt19937 gen(0);
    float avg_game_length = 25; long long game_count = 0;
    long long sims = 0, moves = 0;
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    bool alive = true;
    while (alive) {                                       /* generate.cpp:72-120 */
        int i = 0;
        auto state = env.reset();
        agent.search_probs(state, budget, 0); sims += budget;
        if (elapsed() > seconds) break;
        while (alive) {
            i++;
            auto policy = agent.search_probs(state, budget, 0); sims += budget; moves++;
            int point;
            if (i < avg_game_length * 0.2) {
                auto ptr = policy.data_ptr<float>();
                std::discrete_distribution<> dist(ptr, ptr + policy.numel());
                point = dist(gen);
            } else point = torch::argmax(policy).item<int>();
            torch::Tensor reward; bool done;
            std::tie(state, reward, done) = env.step(state, {point / N, point % N});
            if (elapsed() > seconds) { alive = false; break; }
            if (done) {
                agent.clear();
                game_count++;
                avg_game_length += (i - avg_game_length) / game_count;
                break;
            }
        }
    }
    const double el = elapsed();
    printf("{\"sims\": %lld, \"seconds\": %.3f, \"sims_per_s\": %.3f, \"moves\": %lld, \"games\": %lld, "
           "\"nthreads\": %d, \"torch_threads\": %d, \"N\": %d, \"budget\": %d}\n",
           sims, el, sims / el, moves, game_count, nthreads, at::get_num_threads(), N, budget);
    fflush(stdout);
    _exit(0);
}

/* --------------------------------------------------------------- replayio -- */
/* The trainer's on-disk buffer (replay.hpp:241-274 save, :163-208 load): three
 * tensors through torch::save / torch::load.  replay.hpp itself cannot be
 * compiled here (needs cppzmq), so the two calls it makes are made directly. */
static int cmd_replayio(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "replayio write|read dir\n"); return 2; }
    const std::string mode = argv[2], dir = argv[3];
    if (mode == "write") {
        const int n = 5, N = 3;
        auto s = (torch::arange(n * 3 * N * N) % 3 == 0).to(torch::kUInt8).reshape({n, 3, N, N});
        auto p = (torch::arange(n * N * N, torch::kFloat32) / 100.0).reshape({n, 1, N, N});
        auto r = torch::tensor({1.f, -1.f, -1.f, 1.f, 0.f, 0.f, 1.f, -1.f, 0.f, 0.f}).reshape({n, 2});
        torch::save(s, dir + "/replay.state");
        torch::save(p, dir + "/replay.policy");
        torch::save(r, dir + "/replay.reward");
        return 0;
    }
    torch::Tensor s, p, r;
    torch::load(s, dir + "/replay.state");
    torch::load(p, dir + "/replay.policy");
    torch::load(r, dir + "/replay.reward");
    if (s.dtype() != torch::kUInt8 || p.dtype() != torch::kFloat32 || r.dtype() != torch::kFloat32) return 3;
    printf("%ld %ld %ld %ld | %ld %.6f %.6f\n", (long)s.size(0), (long)s.size(1), (long)p.size(1), (long)r.size(1),
           (long)s.sum().item<int64_t>(), p.sum().item<double>(), r.abs().sum().item<double>());
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: az_ref tower|mcts|bench|replayio ...\n"); return 2; }
    const std::string c = argv[1];
    if (c == "tower") return cmd_tower(argc, argv);
    if (c == "mcts") return cmd_mcts(argc, argv);
    if (c == "bench") return cmd_bench(argc, argv);
    if (c == "replayio") return cmd_replayio(argc, argv);
    return 2;
}
