/* blokus_oracle.cpp -- TEST INFRASTRUCTURE ONLY: a CPU statement of the Blokus environment the library
 * defines (alphazero_amd/csrc/az_blokus_rules.h states the rules in words), written independently of the
 * product: pieces as coordinate lists, orientations through explicit 2x2 integer matrices, legality through
 * neighbour sets on a plain 2-D array -- no bit boards, no shared tables.
 *
 * PARITY UNPINNED: the reference names a Blokus submodule (.gitmodules:7-9) but its directory is empty and its
 * own MCTS cannot drive a four-player game (replay.hpp:127, mcts.hpp:431), so there is no upstream source,
 * fixture or output to pin this against.  What pins it instead: tests/test_blokus_rules.py (piece census against
 * a brute-force polyomino enumeration, hand-made positions, hypothesis playouts against invariants).
 * Interface contract followed: the Env call sites, mcts.hpp:251-271,427; generate.cpp:75,103.
 */
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <set>
#include <utility>
#include <vector>

extern "C" uint32_t azo_philox_u32(uint64_t seed, uint32_t stream, uint32_t counter, uint32_t lane);

namespace {

typedef std::pair<int, int> Cell;                       /* (row, column) */
typedef std::vector<Cell> Shape;

const int N = 20, NP = 4, NPIECES = 21;

/* the 21 pieces by their squares; names as in the rules text */
const std::vector<Shape>& base_pieces() {
    static const std::vector<Shape> P = {
        /* I1 */ {{0, 0}},
        /* I2 */ {{0, 0}, {0, 1}},
        /* I3 */ {{0, 0}, {0, 1}, {0, 2}},
        /* V3 */ {{0, 0}, {1, 0}, {1, 1}},
        /* I4 */ {{0, 0}, {0, 1}, {0, 2}, {0, 3}},
        /* L4 */ {{0, 0}, {1, 0}, {1, 1}, {1, 2}},
        /* T4 */ {{0, 1}, {1, 0}, {1, 1}, {1, 2}},
        /* O4 */ {{0, 0}, {0, 1}, {1, 0}, {1, 1}},
        /* S4 */ {{0, 1}, {0, 2}, {1, 0}, {1, 1}},
        /* F5 */ {{0, 1}, {0, 2}, {1, 0}, {1, 1}, {2, 1}},
        /* I5 */ {{0, 0}, {0, 1}, {0, 2}, {0, 3}, {0, 4}},
        /* L5 */ {{0, 0}, {1, 0}, {1, 1}, {1, 2}, {1, 3}},
        /* N5 */ {{0, 1}, {0, 2}, {0, 3}, {1, 0}, {1, 1}},
        /* P5 */ {{0, 0}, {0, 1}, {1, 0}, {1, 1}, {2, 0}},
        /* T5 */ {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {2, 1}},
        /* U5 */ {{0, 0}, {0, 2}, {1, 0}, {1, 1}, {1, 2}},
        /* V5 */ {{0, 0}, {1, 0}, {2, 0}, {2, 1}, {2, 2}},
        /* W5 */ {{0, 0}, {1, 0}, {1, 1}, {2, 1}, {2, 2}},
        /* X5 */ {{0, 1}, {1, 0}, {1, 1}, {1, 2}, {2, 1}},
        /* Y5 */ {{0, 2}, {1, 0}, {1, 1}, {1, 2}, {1, 3}},
        /* Z5 */ {{0, 0}, {0, 1}, {1, 1}, {2, 1}, {2, 2}},
    };
    return P;
}

Shape normalised(Shape s) {
    int my = s[0].first, mx = s[0].second;
    for (const Cell& c : s) { my = std::min(my, c.first); mx = std::min(mx, c.second); }
    for (Cell& c : s) { c.first -= my; c.second -= mx; }
    std::sort(s.begin(), s.end());
    return s;
}

struct Orient { int piece; Shape cells; int h, w; };

/* identity, three clockwise quarter turns, then the same four after a left-right mirror:
 * (row, col) -> (m[0] row + m[1] col, m[2] row + m[3] col) */
const std::vector<Orient>& orients() {
    static std::vector<Orient> T;
    if (!T.empty()) return T;
    static const int M[8][4] = {{1, 0, 0, 1}, {0, 1, -1, 0}, {-1, 0, 0, -1}, {0, -1, 1, 0},
                                {1, 0, 0, -1}, {0, -1, -1, 0}, {-1, 0, 0, 1}, {0, 1, 1, 0}};
    const std::vector<Shape>& P = base_pieces();
    for (int pc = 0; pc < NPIECES; ++pc) {
        std::vector<Shape> seen;
        for (int t = 0; t < 8; ++t) {
            Shape s;
            for (const Cell& c : P[pc]) s.push_back({M[t][0] * c.first + M[t][1] * c.second, M[t][2] * c.first + M[t][3] * c.second});
            s = normalised(s);
            if (std::find(seen.begin(), seen.end(), s) != seen.end()) continue;
            seen.push_back(s);
            Orient o{pc, s, 0, 0};
            for (const Cell& c : s) { o.h = std::max(o.h, c.first + 1); o.w = std::max(o.w, c.second + 1); }
            T.push_back(o);
        }
    }
    return T;
}

struct Game {
    int board[N][N];                                    /* -1 empty, else owner */
    bool used[NP][NPIECES];
    int last_piece[NP];
    int to_move;
    bool over;
    int moves;
};

void reset(Game& g) {
    for (int y = 0; y < N; ++y) for (int x = 0; x < N; ++x) g.board[y][x] = -1;
    memset(g.used, 0, sizeof g.used);
    for (int p = 0; p < NP; ++p) g.last_piece[p] = -1;
    g.to_move = 0; g.over = false; g.moves = 0;
}

bool on_board(int y, int x) { return y >= 0 && y < N && x >= 0 && x < N; }
int owner(const Game& g, int y, int x) { return on_board(y, x) ? g.board[y][x] : -1; }

bool legal(const Game& g, int p, int action) {
    const std::vector<Orient>& T = orients();
    if (action < 0 || action >= (int)T.size() * N * N) return false;
    const Orient& o = T[action / (N * N)];
    const int y0 = (action % (N * N)) / N, x0 = action % N;
    if (g.used[p][o.piece]) return false;
    bool any_used = false;
    for (int pc = 0; pc < NPIECES; ++pc) any_used = any_used || g.used[p][pc];
    const Cell corner[NP] = {{0, 0}, {0, N - 1}, {N - 1, N - 1}, {N - 1, 0}};
    bool contact = false;
    for (const Cell& c : o.cells) {
        const int y = y0 + c.first, x = x0 + c.second;
        if (!on_board(y, x) || g.board[y][x] != -1) return false;
        if (owner(g, y - 1, x) == p || owner(g, y + 1, x) == p || owner(g, y, x - 1) == p || owner(g, y, x + 1) == p) return false;
        if (!any_used) contact = contact || Cell(y, x) == corner[p];
        else contact = contact || owner(g, y - 1, x - 1) == p || owner(g, y - 1, x + 1) == p ||
                       owner(g, y + 1, x - 1) == p || owner(g, y + 1, x + 1) == p;
    }
    return contact;
}

std::vector<int> legal_actions(const Game& g, int p) {
    std::vector<int> out;
    const int A = (int)orients().size() * N * N;
    for (int a = 0; a < A; ++a) if (legal(g, p, a)) out.push_back(a);
    return out;
}

void scores(const Game& g, int s[NP]) {
    const std::vector<Shape>& P = base_pieces();
    for (int p = 0; p < NP; ++p) {
        int left = 0;
        for (int pc = 0; pc < NPIECES; ++pc) if (!g.used[p][pc]) left += (int)P[pc].size();
        s[p] = left ? -left : 15 + (g.last_piece[p] == 0 ? 5 : 0);
    }
}

/* 1 done, 0 running, -1 illegal */
int step(Game& g, int action, float reward[NP]) {
    for (int p = 0; p < NP; ++p) reward[p] = 0.f;
    if (g.over || !legal(g, g.to_move, action)) return -1;
    const int p = g.to_move;
    const Orient& o = orients()[action / (N * N)];
    const int y0 = (action % (N * N)) / N, x0 = action % N;
    for (const Cell& c : o.cells) g.board[y0 + c.first][x0 + c.second] = p;
    g.used[p][o.piece] = true;
    g.last_piece[p] = o.piece;
    g.moves++;
    for (int k = 1; k <= NP; ++k) {
        const int q = (p + k) % NP;
        if (!legal_actions(g, q).empty()) { g.to_move = q; return 0; }
    }
    g.over = true;
    int s[NP];
    scores(g, s);
    const int best = *std::max_element(s, s + NP);
    const bool all_equal = std::count(s, s + NP, s[0]) == NP;
    for (int q = 0; q < NP; ++q) reward[q] = all_equal ? 0.f : (s[q] == best ? 1.f : -1.f);
    return 1;
}

}  // namespace

extern "C" {

int azo_blokus_orients(void) { return (int)orients().size(); }
/* orientation o: piece, bounding box, cells (row, column) row-major; returns the number of cells */
int azo_blokus_orient(int o, int* piece, int* h, int* w, int8_t* cells_yx) {
    const Orient& t = orients()[(size_t)o];
    *piece = t.piece; *h = t.h; *w = t.w;
    for (size_t i = 0; i < t.cells.size(); ++i) { cells_yx[2 * i] = (int8_t)t.cells[i].first; cells_yx[2 * i + 1] = (int8_t)t.cells[i].second; }
    return (int)t.cells.size();
}

void* azo_blokus_new(void) { Game* g = new Game; reset(*g); return g; }
void azo_blokus_free(void* h) { delete (Game*)h; }
void azo_blokus_reset(void* h) { reset(*(Game*)h); }
int azo_blokus_step(void* h, int action, float* reward) { return step(*(Game*)h, action, reward); }
int azo_blokus_to_move(void* h) { return ((Game*)h)->to_move; }
int azo_blokus_over(void* h) { return ((Game*)h)->over ? 1 : 0; }
/* mask u8 [91*400] (may be null); returns the number of legal actions of the player to move (0 once over) */
int azo_blokus_legal(void* h, uint8_t* mask) {
    const Game& g = *(Game*)h;
    const int A = (int)orients().size() * N * N;
    if (mask) memset(mask, 0, (size_t)A);
    if (g.over) return 0;
    const std::vector<int> acts = legal_actions(g, g.to_move);
    if (mask) for (int a : acts) mask[a] = 1;
    return (int)acts.size();
}
/* cells u8 [400]: 0 empty, 1 + owner */
void azo_blokus_cells(void* h, uint8_t* cells) {
    const Game& g = *(Game*)h;
    for (int y = 0; y < N; ++y) for (int x = 0; x < N; ++x) cells[y * N + x] = (uint8_t)(g.board[y][x] + 1);
}
void azo_blokus_planes(void* h, uint8_t* planes) {
    const Game& g = *(Game*)h;
    for (int y = 0; y < N; ++y)
        for (int x = 0; x < N; ++x) {
            for (int p = 0; p < NP; ++p) planes[p * N * N + y * N + x] = g.board[y][x] == p;
            planes[NP * N * N + y * N + x] = (uint8_t)g.to_move;
        }
}
void azo_blokus_used(void* h, uint32_t* used) {
    const Game& g = *(Game*)h;
    for (int p = 0; p < NP; ++p) { used[p] = 0; for (int pc = 0; pc < NPIECES; ++pc) if (g.used[p][pc]) used[p] |= 1u << pc; }
}
void azo_blokus_scores(void* h, int32_t* s) { int t[NP]; scores(*(Game*)h, t); for (int p = 0; p < NP; ++p) s[p] = t[p]; }

/* One uniformly random legal move, drawn the way the library's playout kernel draws it: the k-th legal
 * action in index order, k = (u * count) >> 32 with u = Philox(seed, stream, counter) word 0.
 * Returns the action played, or -1 when the game is over. */
int azo_blokus_random_move(void* h, uint64_t seed, uint32_t stream, uint32_t counter, float* reward, int* done) {
    Game& g = *(Game*)h;
    if (g.over) { *done = 1; return -1; }
    const std::vector<int> acts = legal_actions(g, g.to_move);
    const uint32_t u = azo_philox_u32(seed, stream, counter, 0);
    const int a = acts[(size_t)(((uint64_t)u * acts.size()) >> 32)];
    *done = step(g, a, reward);
    return a;
}

}  /* extern "C" */
