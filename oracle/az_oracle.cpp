/*
 * az_oracle.cpp -- CPU restatement of the reference self-play hot path.
 * TEST INFRASTRUCTURE ONLY (see az_oracle.h).  Plain C++17, no torch.
 *
 * The arithmetic types of every expression that feeds a visit count follow
 * the reference exactly (int/float/double promotions are spelled out), and
 * this file must be built WITHOUT -ffast-math and WITHOUT -march flags that
 * enable FMA contraction (the Makefile passes -ffp-contract=off).
 */
#include "az_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <numeric>
#include <random>
#include <string>
#include <unordered_map>
#include <vector>

/* ====================================================================== env
 * m,n,k-game family (tictactoe N=3,K=3; gomoku N,K=5).  Contract from the
 * call sites: step is pure and returns (state, reward[2], done)
 * (mcts.hpp:251,261; generate.cpp:103); reward is indexed by absolute player
 * (mcts.hpp:336); possible_actions feeds an ordered map so child order is
 * row-major (mcts.hpp:427-453); reset gives player 0 the move
 * (generate.cpp:75,102).  PARITY UNPINNED: upstream env sources are absent.
 */
extern "C" void azo_env_reset(int N, uint8_t* cells, int* to_move) {
    std::memset(cells, 0, (size_t)N * N);
    *to_move = 0;
}

static bool line_win(int N, int K, const uint8_t* cells, int y, int x, uint8_t stone) {
    static const int DY[4] = {0, 1, 1, 1};
    static const int DX[4] = {1, 0, 1, -1};
    for (int d = 0; d < 4; ++d) {
        int run = 1;
        for (int s = 1; s < K; ++s) {
            int yy = y + s * DY[d], xx = x + s * DX[d];
            if (yy < 0 || yy >= N || xx < 0 || xx >= N || cells[yy * N + xx] != stone) break;
            ++run;
        }
        for (int s = 1; s < K; ++s) {
            int yy = y - s * DY[d], xx = x - s * DX[d];
            if (yy < 0 || yy >= N || xx < 0 || xx >= N || cells[yy * N + xx] != stone) break;
            ++run;
        }
        if (run >= K) return true;
    }
    return false;
}

extern "C" int azo_env_step(int N, int K, uint8_t* cells, int* to_move, int action, float reward[2]) {
    reward[0] = reward[1] = 0.f;
    if (action < 0 || action >= N * N || cells[action] != 0) return -1;
    const int mover = *to_move;
    const uint8_t stone = (uint8_t)(mover + 1);
    cells[action] = stone;
    *to_move = 1 - mover;
    if (line_win(N, K, cells, action / N, action % N, stone)) {
        reward[mover] = 1.f;
        reward[1 - mover] = -1.f;
        return 1;
    }
    for (int i = 0; i < N * N; ++i)
        if (cells[i] == 0) return 0;
    return 1; /* board full: draw, reward {0,0} */
}

extern "C" void azo_env_planes(int N, const uint8_t* cells, int to_move, uint8_t* planes) {
    const int A = N * N;
    for (int i = 0; i < A; ++i) {
        planes[i] = cells[i] == 1;
        planes[A + i] = cells[i] == 2;
        planes[2 * A + i] = (uint8_t)to_move;
    }
}

extern "C" int azo_env_legal(int N, const uint8_t* cells, int32_t* actions) {
    int k = 0;
    for (int i = 0; i < N * N; ++i)
        if (cells[i] == 0) actions[k++] = i;
    return k;
}

/* =============================================================== generators */
static inline uint64_t splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

/* Philox4x32-10 (Salmon et al. 2011): ctr = {counter, stream, 0, 0},
 * key = {seed lo, seed hi}; returns output word lane&3. */
extern "C" uint32_t azo_philox_u32(uint64_t seed, uint32_t stream, uint32_t counter, uint32_t lane) {
    uint32_t c[4] = {counter, stream, 0u, 0u};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return c[lane & 3];
}

/* ================================================================ evaluators
 * Hash pseudo-network: a pure integer function of the planes, then IEEE
 * divisions only, so the HIP twin reproduces it bit for bit.
 */
extern "C" void azo_eval_hash(void*, int N, const uint8_t* planes, float* policy, float* value) {
    const int A = N * N;
    uint64_t H = 0;
    for (int i = 0; i < 3 * A; ++i) H ^= splitmix64((uint64_t)i * 256u + planes[i]);
    uint32_t sum = 0;
    std::vector<uint32_t> w(A);
    for (int i = 0; i < A; ++i) {
        uint64_t r = (splitmix64(H + (uint64_t)i + 1u) >> 40) & 0xFFFFu;
        w[i] = 1u + (uint32_t)((r * r * r) >> 32);
        sum += w[i];
    }
    const float fs = (float)sum;
    for (int i = 0; i < A; ++i) policy[i] = (float)w[i] / fs;
    const int t = (int)((splitmix64(H ^ 0xABCDEFull) >> 40) & 0xFFFFu);
    const float v0 = ((float)t - 32768.0f) / 40000.0f;
    value[0] = v0;
    value[1] = -v0;
}

struct azo_table {
    int N;
    int misses = 0;
    std::unordered_map<std::string, std::pair<std::vector<float>, std::vector<float>>> m;
};
extern "C" azo_table* azo_table_create(int N) { auto* t = new azo_table; t->N = N; return t; }
extern "C" void azo_table_destroy(azo_table* t) { delete t; }
extern "C" void azo_table_add(azo_table* t, const uint8_t* planes, const float* policy, const float* value) {
    const int A = t->N * t->N;
    std::string key((const char*)planes, (size_t)3 * A);
    t->m[key] = {std::vector<float>(policy, policy + A), std::vector<float>(value, value + 2)};
}
extern "C" int azo_table_misses(const azo_table* t) { return t->misses; }
extern "C" void azo_eval_table(void* tp, int N, const uint8_t* planes, float* policy, float* value) {
    auto* t = (azo_table*)tp;
    const int A = N * N;
    auto it = t->m.find(std::string((const char*)planes, (size_t)3 * A));
    if (it == t->m.end()) {
        t->misses++;
        for (int i = 0; i < A; ++i) policy[i] = 1.0f / (float)A;
        value[0] = value[1] = 0.f;
        return;
    }
    std::memcpy(policy, it->second.first.data(), sizeof(float) * A);
    std::memcpy(value, it->second.second.data(), sizeof(float) * 2);
}

/* ===================================================================== tower
 * network.hpp restated in fp32.  Blob order = module registration order
 * (network.hpp:106-114): CBlock, PHead, VHead, RBlock_0.. ; inside a module:
 * conv.weight, conv.bias, then per BatchNorm weight, bias, running_mean,
 * running_var; VHead adds fc1.weight, fc1.bias, fc2.weight, fc2.bias;
 * RBlock_i: conv1.{w,b}, conv2.{w,b}, bn1.{w,b,rm,rv}, bn2.{w,b,rm,rv}.
 */
struct ConvP { int cin, cout, k; const float *w, *b; };
struct BnP { int c; const float *g, *b, *rm, *rv; };
struct azo_tower {
    int N, c_in, c_out, num_res;
    std::vector<int> Cs;
    std::vector<float> blob;
    ConvP stem; BnP stem_bn;
    ConvP pconv;
    ConvP vconv; BnP vbn; const float *fc1w, *fc1b, *fc2w, *fc2b;
    std::vector<ConvP> c1, c2; std::vector<BnP> b1, b2;
};

static size_t conv_sz(int cin, int cout, int k) { return (size_t)cout * cin * k * k + cout; }

extern "C" size_t azo_tower_blob_size(int N, int c_in, int c_out, const int* Cs, int n_Cs) {
    const int num_res = n_Cs - 1;               /* network.hpp:105 */
    const int Ch = Cs[num_res - 1];             /* network.hpp:107-108 */
    size_t s = conv_sz(c_in, Cs[0], 3) + 4 * (size_t)Cs[0];
    s += conv_sz(Ch, c_out, 1);
    s += conv_sz(Ch, 1, 1) + 4 + (size_t)64 * N * N + 64 + 2 * 64 + 2;
    for (int i = 0; i < num_res; ++i)
        s += conv_sz(Cs[i], Cs[i + 1], 3) + conv_sz(Cs[i + 1], Cs[i + 1], 3) + 8 * (size_t)Cs[i + 1];
    return s;
}

extern "C" azo_tower* azo_tower_create(int N, int c_in, int c_out, const int* Cs, int n_Cs,
                                       const float* blob, size_t n_floats) {
    if (n_floats != azo_tower_blob_size(N, c_in, c_out, Cs, n_Cs)) return nullptr;
    auto* t = new azo_tower;
    t->N = N; t->c_in = c_in; t->c_out = c_out; t->num_res = n_Cs - 1;
    t->Cs.assign(Cs, Cs + n_Cs);
    t->blob.assign(blob, blob + n_floats);
    const float* p = t->blob.data();
    auto conv = [&](int cin, int cout, int k) {
        ConvP c{cin, cout, k, p, p + (size_t)cout * cin * k * k};
        p += conv_sz(cin, cout, k);
        return c;
    };
    auto bn = [&](int c) { BnP b{c, p, p + c, p + 2 * c, p + 3 * c}; p += 4 * (size_t)c; return b; };
    const int Ch = Cs[t->num_res - 1];
    t->stem = conv(c_in, Cs[0], 3); t->stem_bn = bn(Cs[0]);
    t->pconv = conv(Ch, c_out, 1);
    t->vconv = conv(Ch, 1, 1); t->vbn = bn(1);
    t->fc1w = p; p += (size_t)64 * N * N; t->fc1b = p; p += 64;
    t->fc2w = p; p += 128; t->fc2b = p; p += 2;
    for (int i = 0; i < t->num_res; ++i) {
        t->c1.push_back(conv(Cs[i], Cs[i + 1], 3));
        t->c2.push_back(conv(Cs[i + 1], Cs[i + 1], 3));
        t->b1.push_back(bn(Cs[i + 1]));
        t->b2.push_back(bn(Cs[i + 1]));
    }
    return t;
}
extern "C" void azo_tower_destroy(azo_tower* t) { delete t; }

/* conv2d, stride 1, zero padding k/2, bias (network.hpp:12,27-28,48,70).  Every output is accumulated in
 * double and rounded to fp32 once: the checker's own summation order then adds nothing, and libtorch's blocked
 * fp32 summation stays within a few ulp of this.  (A sequential fp32 sum over 9 * C terms loses ~1e-5 absolute on an
 * output near 3; in front of a BatchNorm with running_var 1e-4 that is 1e-3 of the normalised activation.) */
static void conv2d(const ConvP& c, int N, const float* x, float* y) {
    const int A = N * N, pad = c.k / 2;
    std::vector<double> acc(A);
    for (int co = 0; co < c.cout; ++co) {
        for (int i = 0; i < A; ++i) acc[i] = c.b[co];
        for (int ci = 0; ci < c.cin; ++ci) {
            const float* xi = x + (size_t)ci * A;
            const float* w = c.w + ((size_t)co * c.cin + ci) * c.k * c.k;
            for (int ky = 0; ky < c.k; ++ky)
                for (int kx = 0; kx < c.k; ++kx) {
                    const double wv = w[ky * c.k + kx];
                    const int dy = ky - pad, dx = kx - pad;
                    const int y0 = std::max(0, -dy), y1 = std::min(N, N - dy);
                    const int x0 = std::max(0, -dx), x1 = std::min(N, N - dx);
                    for (int yy = y0; yy < y1; ++yy) {
                        double* yr = acc.data() + yy * N;
                        const float* xr = xi + (yy + dy) * N + dx;
                        for (int xx = x0; xx < x1; ++xx) yr[xx] += wv * (double)xr[xx];
                    }
                }
        }
        float* yo = y + (size_t)co * A;
        for (int i = 0; i < A; ++i) yo[i] = (float)acc[i];
    }
}
/* BatchNorm2d eval mode, eps 1e-5 (network.hpp:13,29-30,71; evaluator.hpp:145) */
static void bn_eval(const BnP& b, int A, float* x) {
    for (int c = 0; c < b.c; ++c) {
        const float inv = 1.0f / std::sqrt(b.rv[c] + 1e-5f);
        float* xc = x + (size_t)c * A;
        for (int i = 0; i < A; ++i) xc[i] = (xc[i] - b.rm[c]) * inv * b.g[c] + b.b[c];
    }
}
static void relu(float* x, size_t n) { for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.f ? x[i] : 0.f; }

extern "C" void azo_tower_forward(const azo_tower* t, const uint8_t* boards, int B, float* policy, float* value) {
    const int N = t->N, A = N * N;
    int Cmax = t->c_in;
    for (int c : t->Cs) Cmax = std::max(Cmax, c);
    std::vector<float> x((size_t)Cmax * A), h((size_t)Cmax * A), h2((size_t)Cmax * A);
    for (int b = 0; b < B; ++b) {
        for (int i = 0; i < t->c_in * A; ++i) h[i] = (float)boards[(size_t)b * t->c_in * A + i];
        /* ConvBlock (network.hpp:15-17) */
        conv2d(t->stem, N, h.data(), x.data());
        bn_eval(t->stem_bn, A, x.data());
        relu(x.data(), (size_t)t->Cs[0] * A);
        /* ResBlocks (network.hpp:32-38) */
        for (int i = 0; i < t->num_res; ++i) {
            const int C = t->Cs[i + 1];
            conv2d(t->c1[i], N, x.data(), h.data());
            bn_eval(t->b1[i], A, h.data());
            relu(h.data(), (size_t)C * A);
            conv2d(t->c2[i], N, h.data(), h2.data());
            bn_eval(t->b2[i], A, h2.data());
            for (size_t k = 0; k < (size_t)C * A; ++k) h2[k] += x[k];
            relu(h2.data(), (size_t)C * A);
            std::swap(x, h2);
        }
        /* PolicyHead eval: softmax over c_out*N*N (network.hpp:51-60) */
        const int PO = t->c_out * A;
        conv2d(t->pconv, N, x.data(), h.data());
        float mx = h[0];
        for (int i = 1; i < PO; ++i) mx = std::max(mx, h[i]);
        float sum = 0.f;
        for (int i = 0; i < PO; ++i) { h[i] = std::exp(h[i] - mx); sum += h[i]; }
        for (int i = 0; i < PO; ++i) policy[(size_t)b * PO + i] = h[i] / sum;
        /* ValueHead (network.hpp:75-83) */
        conv2d(t->vconv, N, x.data(), h.data());
        bn_eval(t->vbn, A, h.data());
        relu(h.data(), A);
        float f1[64];
        for (int o = 0; o < 64; ++o) {
            float s = t->fc1b[o];
            for (int i = 0; i < A; ++i) s += t->fc1w[(size_t)o * A + i] * h[i];
            f1[o] = s > 0.f ? s : 0.f;
        }
        float z[2];
        for (int o = 0; o < 2; ++o) {
            float s = t->fc2b[o];
            for (int i = 0; i < 64; ++i) s += t->fc2w[o * 64 + i] * f1[i];
            z[o] = s;
        }
        const float m2 = std::max(z[0], z[1]);
        const float e0 = std::exp(z[0] - m2), e1 = std::exp(z[1] - m2);
        value[b * 2 + 0] = e0 / (e0 + e1) * 2.f - 1.f;
        value[b * 2 + 1] = e1 / (e0 + e1) * 2.f - 1.f;
    }
}
extern "C" void azo_eval_tower(void* tp, int, const uint8_t* planes, float* policy, float* value) {
    azo_tower_forward((const azo_tower*)tp, planes, 1, policy, value);
}

/* ================================================================= dirichlet
 * util.h:42-58: gamma(alpha) draws in float, double sum, `f /= sum`. */
static void dirichlet_mt(std::mt19937& rng, int size, float alpha, float* out) {
    std::gamma_distribution<float> dist(alpha);
    for (int i = 0; i < size; ++i) out[i] = dist(rng);
    double sum = 0.;
    for (int i = 0; i < size; ++i) sum += out[i];
    for (int i = 0; i < size; ++i) out[i] = (float)(out[i] / sum);
}
extern "C" void azo_dirichlet(uint64_t* rng_state, int size, float alpha, float* out) {
    std::mt19937 rng((uint32_t)splitmix64(*rng_state));
    ++*rng_state;
    dirichlet_mt(rng, size, alpha, out);
}

/* ====================================================================== mcts
 * Node (node.hpp:9-81): every child stub is a full Node holding the edge
 * statistics p, n, v, q; `add`/`terminal_add` materialise it.
 */
struct Node {
    float p;
    int n = 0;
    float v = 0.f, q = 0.f;
    int player = 0;
    bool terminal = false, expanded = false;
    float reward[2] = {0.f, 0.f};
    std::vector<uint8_t> cells;     /* the node's state (set by add) */
    Node* parent;
    std::vector<std::pair<int, std::unique_ptr<Node>>> children; /* action order */
    Node(float p_, Node* par) : p(p_), parent(par) {}
};

struct azo_mcts {
    int N, K;
    float vl, c_puct;
    azo_eval_fn fn;
    void* user;
    std::unique_ptr<Node> root;
    std::vector<int> root_id;       /* move history of the root (node.hpp:59) */
    std::mt19937 rng{12345u};
    long evals = 0;
    int leaves = 1;                 /* descents per round, see round_k */
    std::vector<uint8_t> planes;
    std::vector<float> policy;
};

extern "C" azo_mcts* azo_mcts_create(int N, int K, float vl, float c_puct, azo_eval_fn fn, void* user) {
    auto* m = new azo_mcts;
    m->N = N; m->K = K; m->vl = vl; m->c_puct = c_puct; m->fn = fn; m->user = user;
    m->planes.resize((size_t)3 * N * N);
    m->policy.resize((size_t)N * N);
    return m;
}
extern "C" void azo_mcts_destroy(azo_mcts* m) { delete m; }
extern "C" void azo_mcts_clear(azo_mcts* m) { m->root.reset(); m->root_id.clear(); }
extern "C" void azo_mcts_seed(azo_mcts* m, uint64_t seed) { m->rng.seed((uint32_t)splitmix64(seed)); }
extern "C" int azo_mcts_root_n(const azo_mcts* m) { return m->root ? m->root->n : 0; }
extern "C" long azo_mcts_eval_count(const azo_mcts* m) { return m->evals; }

/* node.hpp:47-55.  c_puct arrives as int (truncation of the float member);
 * `c_puct * p` is float, sqrt(int) is double, the rest is double, the result
 * narrows to float on return. */
static float ucb(const Node* nd, int c_puct) {
    const Node* par = nd->parent;
    const float cp = (float)c_puct * nd->p;
    const double s = std::sqrt((double)(par->n - 1));
    const double u = (double)cp * s / (double)(1 + nd->n);
    return (float)((double)nd->q + u);
}

/* mcts.hpp:350-381 with the tie rule "first in map order" */
static size_t choose(const Node* node, float c_puct) {
    float max_val = -100.f;
    size_t best = (size_t)-1;
    const int c = (int)c_puct; /* node.hpp:47 takes int */
    for (size_t k = 0; k < node->children.size(); ++k) {
        const float val = ucb(node->children[k].second.get(), c);
        if (val > max_val) { max_val = val; best = k; }
    }
    return best;
}

/* mcts.hpp:425-455 */
static void append_children(azo_mcts* m, Node* node, const float* policy, const float* noise) {
    const int A = m->N * m->N;
    std::vector<int32_t> acts(A);
    const int na = azo_env_legal(m->N, node->cells.data(), acts.data());
    std::vector<float> gen;
    if (!node->parent && !noise) {
        gen.resize(na);
        dirichlet_mt(m->rng, na, 0.05f, gen.data());
        noise = gen.data();
    }
    node->children.reserve(na);
    for (int k = 0; k < na; ++k) {
        float prior = policy[acts[k]];
        if (!node->parent) prior = (float)(0.75 * (double)prior + 0.25 * (double)noise[k]);
        node->children.emplace_back(acts[k], std::make_unique<Node>(prior, node));
    }
}

static void evaluate(azo_mcts* m, const uint8_t* cells, int to_move, float* value) {
    azo_env_planes(m->N, cells, to_move, m->planes.data());
    m->fn(m->user, m->N, m->planes.data(), m->policy.data(), value);
    m->evals++;
}

/* mcts.hpp:183-346 at nthreads = 1 */
static void simulate_once(azo_mcts* m) {
    Node* focus = m->root.get();
    int action = -1;
    while (!focus->terminal && !focus->children.empty()) {          /* :217 */
        const size_t k = choose(focus, m->c_puct);
        action = focus->children[k].first;
        focus = focus->children[k].second.get();
        focus->n = (int)((float)focus->n + m->vl);                  /* :225 int += float */
        focus->v -= m->vl;                                          /* :226 */
        focus->q = focus->v / (float)focus->n;                      /* :227 */
    }
    float result[2];
    if (focus->terminal) {                                          /* :232-233 */
        result[0] = focus->reward[0]; result[1] = focus->reward[1];
    } else if (focus->parent) {                                     /* _eval :240-321 */
        Node* parent = focus->parent;
        std::vector<uint8_t> cells = parent->cells;
        int to_move = parent->player;
        float rew[2];
        const int done = azo_env_step(m->N, m->K, cells.data(), &to_move, action, rew);
        if (done) {                                                 /* :266-269 */
            focus->cells = cells; focus->player = to_move; focus->terminal = true;
            focus->reward[0] = rew[0]; focus->reward[1] = rew[1];
            result[0] = rew[0]; result[1] = rew[1];
        } else {
            evaluate(m, cells.data(), to_move, result);             /* :306-314 */
            focus->cells = cells; focus->player = to_move; focus->expanded = true; /* :316 */
            append_children(m, focus, m->policy.data(), nullptr);   /* :318 */
        }
    } else {
        /* a root with no children that is not terminal cannot be searched in
         * the reference (uninitialised iterator, mcts.hpp:214,235) */
        result[0] = result[1] = 0.f;
    }
    /* __backup :323-346 */
    Node* node = focus;
    while (node->parent) {
        const int player = node->parent->player;
        const float value = result[player];
        node->n = (int)((float)node->n + (1.f - m->vl));            /* :337 */
        node->v += value + m->vl;                                   /* :338 */
        node->q = node->v / (float)node->n;                         /* :339 */
        node = node->parent;
    }
    node->n++;                                                      /* :345 */
}

/* ------------------------------------------------------------------------
 * K leaves per tree per round (the reference's nthreads = K with virtual loss,
 * mcts.hpp:54-72,216-231, whose interleaving is racy) given ONE defined order:
 * a round performs up to K descents in sequence, each leaving its virtual loss
 * in place, then evaluates the collected leaves, then expands and backs them
 * up in the same order.  A descent that reaches a stub already waiting for its
 * evaluation in this round (the reference thread would block on that node's
 * lock, mcts.hpp:224) is undone and ends the round's descents.  K = 1 is
 * exactly simulate_once.
 */
struct PendK { Node* leaf; int kind; std::vector<uint8_t> cells; int to_move; float result[2]; };

static int round_k(azo_mcts* m, int K, int remaining) {
    std::vector<PendK> pend;
    for (int k = 0; k < K && (int)pend.size() < remaining; ++k) {
        Node* focus = m->root.get();
        int action = -1;
        struct Old { Node* nd; int n; float v, q; };
        std::vector<Old> path;
        while (!focus->terminal && !focus->children.empty()) {
            const size_t c = choose(focus, m->c_puct);
            action = focus->children[c].first;
            focus = focus->children[c].second.get();
            path.push_back({focus, focus->n, focus->v, focus->q});
            focus->n = (int)((float)focus->n + m->vl);
            focus->v -= m->vl;
            focus->q = focus->v / (float)focus->n;
        }
        PendK pk{focus, 0, {}, 0, {0.f, 0.f}};
        if (focus->terminal) {
            pk.kind = 3; pk.result[0] = focus->reward[0]; pk.result[1] = focus->reward[1];
        } else if (focus->parent) {
            bool collide = false;
            for (auto& q : pend) collide = collide || (q.leaf == focus && q.kind == 2);
            if (collide) {                       // undo this descent exactly and stop
                for (auto it = path.rbegin(); it != path.rend(); ++it) { it->nd->n = it->n; it->nd->v = it->v; it->nd->q = it->q; }
                break;
            }
            pk.cells = focus->parent->cells;
            pk.to_move = focus->parent->player;
            float rew[2];
            const int done = azo_env_step(m->N, m->K, pk.cells.data(), &pk.to_move, action, rew);
            if (done) {
                focus->cells = pk.cells; focus->player = pk.to_move; focus->terminal = true;
                focus->reward[0] = rew[0]; focus->reward[1] = rew[1];
                pk.kind = 3; pk.result[0] = rew[0]; pk.result[1] = rew[1];
            } else pk.kind = 2;
        } else {
            pk.kind = 3;                         // unsearchable root, see simulate_once
        }
        pend.push_back(pk);
    }
    std::vector<std::vector<float>> pols(pend.size());
    for (size_t i = 0; i < pend.size(); ++i)
        if (pend[i].kind == 2) {
            evaluate(m, pend[i].cells.data(), pend[i].to_move, pend[i].result);
            pols[i] = m->policy;
        }
    for (size_t i = 0; i < pend.size(); ++i) {
        Node* leaf = pend[i].leaf;
        if (pend[i].kind == 2) {
            leaf->cells = pend[i].cells; leaf->player = pend[i].to_move; leaf->expanded = true;
            append_children(m, leaf, pols[i].data(), nullptr);
        }
        Node* node = leaf;
        while (node->parent) {
            const float value = pend[i].result[node->parent->player];
            node->n = (int)((float)node->n + (1.f - m->vl));
            node->v += value + m->vl;
            node->q = node->v / (float)node->n;
            node = node->parent;
        }
        node->n++;
    }
    return (int)pend.size();
}

/* node.hpp:83-102 + mcts.hpp:385-421 */
static void make_root(azo_mcts* m, const std::vector<int>& id, const uint8_t* cells, int to_move,
                      const float* noise) {
    Node* target = nullptr;
    Node* target_parent = nullptr;
    size_t target_slot = 0;
    if (m->root && id.size() >= m->root_id.size() &&
        std::equal(m->root_id.begin(), m->root_id.end(), id.begin())) {
        target = m->root.get();
        for (size_t idx = m->root_id.size(); idx < id.size() && target; ++idx) {
            Node* next = nullptr;
            for (size_t k = 0; k < target->children.size(); ++k)
                if (target->children[k].first == id[idx]) {
                    next = target->children[k].second.get();
                    target_parent = target; target_slot = k;
                    break;
                }
            target = next;
        }
    }
    /* A stub that was never visited has no state; searching from it is
     * undefined in the reference.  Treated here (and in the HIP path) as a miss. */
    if (target && target != m->root.get() && !target->expanded && !target->terminal) target = nullptr;
    if (!target) {                                                  /* :392-414 */
        m->root = std::make_unique<Node>(1.f, nullptr);
        m->root_id = id;
        float value[2];
        evaluate(m, cells, to_move, value);
        Node* r = m->root.get();
        r->cells.assign(cells, cells + (size_t)m->N * m->N);
        r->player = to_move; r->expanded = true;
        r->n++;
        append_children(m, r, m->policy.data(), noise);
    } else if (target != m->root.get()) {                           /* :417-420 */
        std::unique_ptr<Node> keep = std::move(target_parent->children[target_slot].second);
        keep->parent = nullptr;   /* the old root dies, weak_ptr expires */
        m->root = std::move(keep);
        m->root_id = id;
    }
}

extern "C" int azo_mcts_search(azo_mcts* m, const int32_t* history, int hist_len, int budget,
                               const float* noise, float* out_policy) {
    const int A = m->N * m->N;
    std::vector<uint8_t> cells(A);
    int to_move;
    azo_env_reset(m->N, cells.data(), &to_move);
    std::vector<int> id(history, history + hist_len);
    bool over = false;
    for (int i = 0; i < hist_len; ++i) {
        float r[2];
        if (over) return -1;
        const int d = azo_env_step(m->N, m->K, cells.data(), &to_move, history[i], r);
        if (d < 0) return -1;
        over = d == 1;
    }
    if (over) return -1;
    make_root(m, id, cells.data(), to_move, noise);
    if (m->leaves > 1) {
        const int target = budget > 1 ? budget : 1;
        int done = 0;
        while (done < target) done += round_k(m, m->leaves, target - done);
    } else {
        int count = 0;
        do { simulate_once(m); } while (!(++count > budget - 1));  /* :161-174, nthreads=1 */
    }
    for (int i = 0; i < A; ++i) out_policy[i] = 0.f;               /* :124 */
    const Node* r = m->root.get();
    for (const auto& ch : r->children)                              /* :136-146 */
        out_policy[ch.first] = (float)ch.second->n / (float)r->n;
    return 0;
}

extern "C" void azo_mcts_set_leaves(azo_mcts* m, int K) { m->leaves = K > 1 ? K : 1; }

extern "C" int azo_mcts_search_k(azo_mcts* m, const int32_t* history, int hist_len, int budget, int K,
                                 const float* noise, float* out_policy) {
    const int keep = m->leaves;
    azo_mcts_set_leaves(m, K);
    const int rc = azo_mcts_search(m, history, hist_len, budget, noise, out_policy);
    m->leaves = keep;
    return rc;
}

extern "C" int azo_mcts_dump(const azo_mcts* m, int cap, int32_t* parent, int32_t* action,
                             int32_t* n, float* v, float* p, int32_t* kind) {
    if (!m->root) return 0;
    std::vector<const Node*> q{m->root.get()};
    int e = 0;
    for (size_t qi = 0; qi < q.size(); ++qi) {
        const Node* nd = q[qi];
        for (const auto& ch : nd->children) {
            const Node* c = ch.second.get();
            int kd = 0;
            if (c->terminal) kd = c->reward[0] > 0 ? 2 : (c->reward[1] > 0 ? 3 : 4);
            else if (c->expanded) kd = 1;
            if (e < cap) {
                parent[e] = (int32_t)qi; action[e] = ch.first; n[e] = c->n;
                v[e] = c->v; p[e] = c->p; kind[e] = kd;
            }
            ++e;
            if (kd == 1) q.push_back(c);
        }
    }
    return e;
}

/* ================================================================= self-play
 * generate.cpp:72-120 for one game.  Sampling draws from the visit counts
 * with the counter-based generator (the reference seeds mt19937 from
 * random_device, so only the distribution is defined there). */
extern "C" int azo_selfplay_game(azo_mcts* m, int budget, uint64_t seed, uint32_t slot,
                                 const float* noise, float* avg_len, int64_t* game_count,
                                 uint32_t* sample_counter,
                                 int max_moves, uint8_t* cells_out, uint8_t* tomove_out,
                                 float* policy_out, int32_t* actions_out, azo_game_result* res) {
    const int N = m->N, A = N * N;
    std::vector<uint8_t> cells(A);
    int to_move;
    azo_env_reset(N, cells.data(), &to_move);                       /* :75 */
    std::vector<int32_t> hist;
    std::vector<float> policy(A);
    azo_mcts_clear(m);
    azo_mcts_search(m, hist.data(), 0, budget, noise, policy.data()); /* :77 warm-up */
    int i = 0;
    for (;;) {
        i++;                                                        /* :82 */
        azo_mcts_search(m, hist.data(), (int)hist.size(), budget, noise, policy.data()); /* :84 */
        if (i > max_moves) return -1;
        std::memcpy(cells_out + (size_t)(i - 1) * A, cells.data(), A);   /* :85 */
        tomove_out[i - 1] = (uint8_t)to_move;
        std::memcpy(policy_out + (size_t)(i - 1) * A, policy.data(), sizeof(float) * A);
        int point = 0;
        const Node* r = m->root.get();
        if ((double)i < (double)*avg_len * 0.2) {                   /* :88 */
            const uint32_t u = azo_philox_u32(seed, slot, (*sample_counter)++, 0);
            uint32_t total = 0;
            for (const auto& ch : r->children) total += (uint32_t)ch.second->n;
            const uint32_t pick = (uint32_t)(((uint64_t)u * total) >> 32);
            uint32_t acc = 0;
            for (const auto& ch : r->children) {
                acc += (uint32_t)ch.second->n;
                if (acc > pick) { point = ch.first; break; }
            }
        } else {                                                    /* :95 first max */
            float best = policy[0];
            for (int k = 1; k < A; ++k)
                if (policy[k] > best) { best = policy[k]; point = k; }
        }
        actions_out[i - 1] = point;
        float rew[2];
        const int done = azo_env_step(N, m->K, cells.data(), &to_move, point, rew); /* :103 */
        hist.push_back(point);
        if (done) {                                                 /* :110-118 */
            azo_mcts_clear(m);
            (*game_count)++;
            *avg_len += ((float)i - *avg_len) / (float)*game_count;  /* :28 */
            res->n_moves = i; res->reward[0] = rew[0]; res->reward[1] = rew[1];
            return i;
        }
    }
}

/* replay.hpp:80-112: for r in 0..3: rot90(r,{1,2}) then the same .flip(1).
 * rot90(1): out[i][j] = x[j][N-1-i]; rot90(2): x[N-1-i][N-1-j];
 * rot90(3): x[N-1-j][i]; flip(1): out[i][j] = in[N-1-i][j]. */
template <class T>
static void sym_plane(int N, int r, bool flip, const T* x, T* out) {
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) {
            const int ii = flip ? N - 1 - i : i;
            int sy, sx;
            switch (r) {
                case 0: sy = ii; sx = j; break;
                case 1: sy = j; sx = N - 1 - ii; break;
                case 2: sy = N - 1 - ii; sx = N - 1 - j; break;
                default: sy = N - 1 - j; sx = ii; break;
            }
            out[i * N + j] = x[sy * N + sx];
        }
}
extern "C" void azo_replay_symmetries(int N, int C, const uint8_t* state, const float* policy,
                                      uint8_t* states_out, float* policies_out) {
    const int A = N * N;
    int v = 0;
    for (int r = 0; r < 4; ++r)
        for (int f = 0; f < 2; ++f, ++v) {
            for (int c = 0; c < C; ++c)
                sym_plane<uint8_t>(N, r, f, state + (size_t)c * A, states_out + ((size_t)v * C + c) * A);
            sym_plane<float>(N, r, f, policy, policies_out + (size_t)v * A);
        }
}
