// az_blokus_rules.h -- Blokus (20x20, four players, 21 polyominoes each) as this library defines it, shared
// by the device kernels (blokus.hip), the C ABI's host twin and the C++ Env shim.  The reference names a Blokus
// environment (.gitmodules:7-9) but ships no source for it and its MCTS cannot drive it (reward length 2,
// replay.hpp:127; 2-D policy accessor, mcts.hpp:431), so the rules below are this build's own statement of
// the standard game, written against the Env concept's call sites (mcts.hpp:251-271,427; generate.cpp:75,103):
//
//   board      20 x 20 cells, 0 = empty, 1 + p = a square of player p (p = 0..3)
//   pieces     21 per player: I1 I2 I3 V3 I4 L4 T4 O4 S4 F5 I5 L5 N5 P5 T5 U5 V5 W5 X5 Y5 Z5 (89 squares)
//   action     (o, y, x): orientation o in [0, 91) of some piece with the top-left corner of its bounding box
//              at row y, column x; index = o * 400 + y * 20 + x, so a policy is [91, 20, 20] = get_action_shape()
//              with c_out = 91 and the children of a node are in (o, y, x) order like every other Env here
//   orientations  per piece in the order above, the eight transforms identity, rot90, rot180, rot270 (clockwise),
//              mirror (left-right), mirror+rot90, mirror+rot180, mirror+rot270, each normalised to its bounding
//              box, duplicates dropped (first occurrence kept): 91 in all = the fixed polyominoes of 1..5 cells
//   legal      piece unused; all cells on the board and empty; no cell edge-adjacent to an own square; and
//              either (first piece) it covers the player's corner -- player 0 (0,0), 1 (0,19), 2 (19,19),
//              3 (19,0) -- or (later) some cell is diagonally adjacent to an own square
//   turn order 0,1,2,3 cyclically; a player without a legal move is skipped (for good: legality never comes
//              back); the game ends when nobody can move
//   score      -(squares left in hand); +15 with nothing left, and +5 more if the last piece was I1
//   reward     f32[4] by absolute player at the end: +1 for the top score(s), -1 for the others, all 0 if
//              all four tie; zero before the end
//   planes     u8 [5,20,20]: squares of player 0..3, then a plane filled with the player to move
#pragma once
#include <stdint.h>
#include <string.h>

#define AZB_N 20
#define AZB_CELLS 400
#define AZB_PLAYERS 4
#define AZB_PIECES 21
#define AZB_ORIENTS 91
#define AZB_ACTIONS (AZB_ORIENTS * AZB_CELLS)
#define AZB_PLANES 5

struct AzbOrient {
    uint8_t piece, h, w, n;         // piece index, bounding box, number of cells
    uint8_t rows[5];                // rows[r] bit c = cell (r, c) of the bounding box
    int8_t cy[5], cx[5];            // the cells, row-major
};

// the 21 pieces, each as rows of a small picture ('#' = square)
static const char* const AZB_SHAPES[AZB_PIECES][5] = {
    {"#"}, {"##"}, {"###"}, {"#.", "##"}, {"####"}, {"#..", "###"}, {".#.", "###"}, {"##", "##"}, {".##", "##."},
    {".##", "##.", ".#."}, {"#####"}, {"#...", "####"}, {".###", "##.."}, {"##", "##", "#."}, {"###", ".#.", ".#."},
    {"#.#", "###"}, {"#..", "#..", "###"}, {"#..", "##.", ".##"}, {".#.", "###", ".#."}, {"..#.", "####"},
    {"##.", ".#.", ".##"},
};

// Fills table[91]; returns the number of orientations found (91).
static inline int azb_build_orients(AzbOrient* table) {
    int count = 0;
    for (int pc = 0; pc < AZB_PIECES; ++pc) {
        int by[5], bx[5], n = 0;
        for (int r = 0; r < 5 && AZB_SHAPES[pc][r]; ++r)
            for (int c = 0; AZB_SHAPES[pc][r][c]; ++c)
                if (AZB_SHAPES[pc][r][c] == '#') { by[n] = r; bx[n] = c; ++n; }
        const int first = count;
        for (int t = 0; t < 8; ++t) {
            int y[5] = {0, 0, 0, 0, 0}, x[5] = {0, 0, 0, 0, 0};
            for (int i = 0; i < n; ++i) {
                int yy = by[i], xx = t >= 4 ? -bx[i] : bx[i];              // mirror left-right first
                for (int q = 0; q < (t & 3); ++q) { const int ny = xx, nx = -yy; yy = ny; xx = nx; }   // rot90 clockwise
                y[i] = yy; x[i] = xx;
            }
            int my = y[0], mx = x[0];
            for (int i = 1; i < n; ++i) { if (y[i] < my) my = y[i]; if (x[i] < mx) mx = x[i]; }
            AzbOrient o;
            memset(&o, 0, sizeof o);
            o.piece = (uint8_t)pc; o.n = (uint8_t)n;
            for (int i = 0; i < n; ++i) {
                const int yy = y[i] - my, xx = x[i] - mx;
                o.rows[yy] |= (uint8_t)(1u << xx);
                if (yy + 1 > o.h) o.h = (uint8_t)(yy + 1);
                if (xx + 1 > o.w) o.w = (uint8_t)(xx + 1);
            }
            int k = 0;
            for (int r = 0; r < o.h; ++r)
                for (int c = 0; c < o.w; ++c)
                    if (o.rows[r] >> c & 1) { o.cy[k] = (int8_t)r; o.cx[k] = (int8_t)c; ++k; }
            bool dup = false;
            for (int j = first; j < count && !dup; ++j) dup = table[j].h == o.h && table[j].w == o.w && memcmp(table[j].rows, o.rows, 5) == 0;
            if (!dup) table[count++] = o;
        }
    }
    return count;
}

// One game on the host: the same rules as the kernels, cell by cell.
struct AzbGame {
    uint8_t cells[AZB_CELLS];       // 0 empty, 1 + owner
    uint32_t used[AZB_PLAYERS];     // bit pc: piece pc has been played
    uint8_t last_piece[AZB_PLAYERS];
    int to_move;                    // player to move; meaningless once over
    int over;
};

static inline void azb_reset(AzbGame* g) { memset(g, 0, sizeof *g); }

static inline bool azb_legal(const AzbGame* g, const AzbOrient* T, int p, int action) {
    if (action < 0 || action >= AZB_ACTIONS) return false;
    const AzbOrient& o = T[action / AZB_CELLS];
    const int y0 = action % AZB_CELLS / AZB_N, x0 = action % AZB_N;
    if (g->used[p] >> o.piece & 1u) return false;
    if (y0 + o.h > AZB_N || x0 + o.w > AZB_N) return false;
    static const int CY[4] = {0, 0, AZB_N - 1, AZB_N - 1}, CX[4] = {0, AZB_N - 1, AZB_N - 1, 0};
    const uint8_t mine = (uint8_t)(p + 1);
    bool touch = false;
    for (int i = 0; i < o.n; ++i) {
        const int y = y0 + o.cy[i], x = x0 + o.cx[i];
        if (g->cells[y * AZB_N + x]) return false;
        if ((y > 0 && g->cells[(y - 1) * AZB_N + x] == mine) || (y < AZB_N - 1 && g->cells[(y + 1) * AZB_N + x] == mine) ||
            (x > 0 && g->cells[y * AZB_N + x - 1] == mine) || (x < AZB_N - 1 && g->cells[y * AZB_N + x + 1] == mine)) return false;
        if (g->used[p] == 0) touch = touch || (y == CY[p] && x == CX[p]);
        else
            for (int d = 0; d < 4; ++d) {
                const int yy = y + (d & 1 ? 1 : -1), xx = x + (d & 2 ? 1 : -1);
                if (yy >= 0 && yy < AZB_N && xx >= 0 && xx < AZB_N && g->cells[yy * AZB_N + xx] == mine) touch = true;
            }
    }
    return touch;
}

static inline bool azb_can_move(const AzbGame* g, const AzbOrient* T, int p) {
    for (int a = 0; a < AZB_ACTIONS; ++a) if (azb_legal(g, T, p, a)) return true;
    return false;
}

static inline void azb_scores(const AzbGame* g, const AzbOrient* T, int score[AZB_PLAYERS]) {
    int size[AZB_PIECES];
    for (int o = 0; o < AZB_ORIENTS; ++o) size[T[o].piece] = T[o].n;
    for (int p = 0; p < AZB_PLAYERS; ++p) {
        int left = 0;
        for (int pc = 0; pc < AZB_PIECES; ++pc) if (!(g->used[p] >> pc & 1u)) left += size[pc];
        score[p] = -left;
        if (left == 0) score[p] = 15 + (g->last_piece[p] == 0 ? 5 : 0);
    }
}

static inline void azb_final_reward(const int score[AZB_PLAYERS], float reward[AZB_PLAYERS]) {
    int best = score[0];
    bool all_equal = true;
    for (int p = 1; p < AZB_PLAYERS; ++p) { if (score[p] > best) best = score[p]; all_equal = all_equal && score[p] == score[0]; }
    for (int p = 0; p < AZB_PLAYERS; ++p) reward[p] = all_equal ? 0.f : (score[p] == best ? 1.f : -1.f);
}

// returns 1 done, 0 running, -1 illegal (state untouched); reward by absolute player
static inline int azb_step(AzbGame* g, const AzbOrient* T, int action, float reward[AZB_PLAYERS]) {
    for (int p = 0; p < AZB_PLAYERS; ++p) reward[p] = 0.f;
    if (g->over || !azb_legal(g, T, g->to_move, action)) return -1;
    const int p = g->to_move;
    const AzbOrient& o = T[action / AZB_CELLS];
    const int y0 = action % AZB_CELLS / AZB_N, x0 = action % AZB_N;
    for (int i = 0; i < o.n; ++i) g->cells[(y0 + o.cy[i]) * AZB_N + x0 + o.cx[i]] = (uint8_t)(p + 1);
    g->used[p] |= 1u << o.piece;
    g->last_piece[p] = o.piece;
    for (int k = 1; k <= AZB_PLAYERS; ++k) {
        const int q = (p + k) % AZB_PLAYERS;
        if (azb_can_move(g, T, q)) { g->to_move = q; return 0; }
    }
    g->over = 1;
    int score[AZB_PLAYERS];
    azb_scores(g, T, score);
    azb_final_reward(score, reward);
    return 1;
}

static inline void azb_planes(const AzbGame* g, uint8_t* planes) {
    for (int i = 0; i < AZB_CELLS; ++i) {
        for (int p = 0; p < AZB_PLAYERS; ++p) planes[p * AZB_CELLS + i] = g->cells[i] == p + 1;
        planes[AZB_PLAYERS * AZB_CELLS + i] = (uint8_t)g->to_move;
    }
}
