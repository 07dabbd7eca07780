// Game rules for the host-side search: tic-tac-toe and Hex, restated from the reference
// (engine/src/ttt/core.rs:13-277, engine/src/hex/core.rs:5-371, traits engine/src/game/mod.rs:8-107).
// Chess lives in chess.h.
//
// Every game type G provides
//   G::BOARD, G::MOVES, G::PLANES, G::PLANE_WORDS, G::REPETITION_LIMIT (0 = none)
//   G::Move      { flipped(), nn_idx(), operator== }
//   G::Position  { initial(), turn(), legal_moves(vec&), moved(Move), status(), flipped(),
//                  planes(u64*), hash(), operator== }
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace cattus {

enum Color : uint8_t { PLAYER1 = 0, PLAYER2 = 1 };
inline Color opposite(Color c) { return c == PLAYER1 ? PLAYER2 : PLAYER1; }

// GameStatus (game/mod.rs:86-98): ongoing, or finished with winner +1 (Player1) / -1 (Player2) / 0 (draw),
// the value GameColor::to_signed_one gives (game/mod.rs:76-82).
struct Status {
    bool finished;
    int8_t winner;
    static Status ongoing() { return {false, 0}; }
    static Status won(Color c) { return {true, (int8_t)(c == PLAYER1 ? 1 : -1)}; }
    static Status draw() { return {true, 0}; }
};

inline uint64_t mix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// ------------------------------------------------------------------------------------ ttt
struct TttGame {
    static constexpr int BOARD = 3, MOVES = 9, PLANES = 3, PLANE_WORDS = 1, REPETITION_LIMIT = 0;
    static const char* name() { return "tictactoe"; }

    struct Move {
        uint8_t idx;
        Move flipped() const { return *this; }          // ttt/core.rs:44-46
        int nn_idx() const { return idx; }               // ttt/core.rs:48-50
        bool operator==(const Move& o) const { return idx == o.idx; }
        std::string str() const { return "(" + std::to_string(idx / 3) + ", " + std::to_string(idx % 3) + ")"; }
    };

    struct Position {
        uint16_t x = 0, o = 0;
        Color to_move = PLAYER1;
        int8_t winner = 0;  // 0 none, +1 Player1 (x), -1 Player2 (o)

        static Position initial() { return Position(); }
        Color turn() const { return to_move; }

        void check_winner() {  // ttt/core.rs:166-190: x is tested before o for each line
            static const uint16_t lines[8] = {0b111000000, 0b000111000, 0b000000111, 0b100100100,
                                              0b010010010, 0b001001001, 0b100010001, 0b001010100};
            for (uint16_t l : lines) {
                if ((x & l) == l) { winner = 1; return; }
                if ((o & l) == l) { winner = -1; return; }
            }
            winner = 0;
        }
        Status status() const {  // ttt/core.rs:227-235
            if (winner) return {true, winner};
            if ((x | o) == 0x1ff) return Status::draw();
            return Status::ongoing();
        }
        void legal_moves(std::vector<Move>& out) const {  // row-major empty cells, ttt/core.rs:208-218
            out.clear();
            for (int i = 0; i < 9; i++)
                if (!((x | o) >> i & 1)) out.push_back(Move{(uint8_t)i});
        }
        Position moved(Move m) const {  // ttt/core.rs:142-155
            Position r = *this;
            (to_move == PLAYER1 ? r.x : r.o) |= (uint16_t)(1u << m.idx);
            r.to_move = opposite(to_move);
            r.check_winner();
            return r;
        }
        Position flipped() const {  // ttt/core.rs:237-244
            Position r;
            r.x = o, r.o = x, r.to_move = opposite(to_move), r.winner = (int8_t)-winner;
            return r;
        }
        void planes(uint64_t* out) const {  // ttt/net.rs:14-24
            out[0] = x, out[1] = o, out[2] = 0x1ff;
        }
        bool operator==(const Position& p) const { return x == p.x && o == p.o && to_move == p.to_move && winner == p.winner; }
        uint64_t hash() const { return mix64(((uint64_t)x << 32) | ((uint64_t)o << 16) | ((uint64_t)to_move << 8) | (uint8_t)winner); }
    };
};

// ------------------------------------------------------------------------------------ hex
typedef unsigned __int128 u128;

template <int N>
struct HexGame {
    static_assert(N >= 2 && N <= 11, "hex board size");
    static constexpr int BOARD = N, MOVES = N * N, PLANES = 3, PLANE_WORDS = 2, REPETITION_LIMIT = 0;
    static const char* name() {
        static const std::string s = "hex" + std::to_string(N);
        return s.c_str();
    }

    struct Move {
        uint8_t idx;
        Move flipped() const { return Move{(uint8_t)((idx % N) * N + idx / N)}; }  // hex/core.rs:36-38
        int nn_idx() const { return idx; }
        bool operator==(const Move& o) const { return idx == o.idx; }
        std::string str() const { return "(" + std::to_string(idx / N) + ", " + std::to_string(idx % N) + ")"; }
    };

    static u128 bit(int i) { return (u128)1 << i; }
    static u128 transpose(u128 b) {  // HexBitboard::flip, hex/core.rs:61-71
        u128 f = 0;
        for (int r = 0; r < N; r++)
            for (int c = 0; c < N; c++)
                if (b >> (r * N + c) & 1) f |= bit(c * N + r);
        return f;
    }

    struct Position {
        u128 red = 0, blue = 0;
        u128 left_red_reach = 0, top_blue_reach = 0;
        Color to_move = PLAYER1;
        uint8_t empties = N * N;
        int8_t winner = 0;

        static Position initial() { return Position(); }
        // HexPosition::new_from_board (hex/core.rs:143-174)
        static Position from_board(u128 red, u128 blue, Color turn) {
            Position s;
            s.red = red, s.blue = blue, s.to_move = turn;
            for (int r = 0; r < N; r++)
                for (int c = 0; c < N; c++) {
                    const int idx = r * N + c;
                    const bool is_red = red >> idx & 1, is_blue = !is_red && (blue >> idx & 1);
                    if (!is_red && !is_blue) continue;
                    s.empties--;
                    const Color col = is_red ? PLAYER1 : PLAYER2;
                    if (col == PLAYER1 ? c == 0 : r == 0) s.update_reach(r, c, col);
                }
            return s;
        }
        Color turn() const { return to_move; }

        template <typename F>
        static void foreach_neighbor(int r, int c, F&& op) {  // hex/core.rs:206-216
            static const int dirs[6][2] = {{0, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, 0}, {1, 1}};
            for (auto& d : dirs) {
                const int nr = r + d[0], nc = c + d[1];
                if (nr < 0 || nr >= N || nc < 0 || nc >= N) continue;
                op(nr, nc);
            }
        }
        void update_reach(int r, int c, Color player) {  // hex/core.rs:218-270
            const u128 board = player == PLAYER1 ? red : blue;
            u128& reach = player == PLAYER1 ? left_red_reach : top_blue_reach;
            u128 layer = 0;
            bool upd = player == PLAYER1 ? c == 0 : r == 0;
            foreach_neighbor(r, c, [&](int nr, int nc) { upd = upd || (reach >> (nr * N + nc) & 1); });
            if (upd) {
                reach |= bit(r * N + c);
                layer |= bit(r * N + c);
            }
            while (layer) {
                const uint64_t lo = (uint64_t)layer;
                const int idx = lo ? __builtin_ctzll(lo) : 64 + __builtin_ctzll((uint64_t)(layer >> 64));
                layer &= ~bit(idx);
                const int rr = idx / N, cc = idx % N;
                if (player == PLAYER1 ? cc == N - 1 : rr == N - 1) {
                    winner = player == PLAYER1 ? 1 : -1;
                } else {
                    foreach_neighbor(rr, cc, [&](int nr, int nc) {
                        const int n = nr * N + nc;
                        if (!(reach >> n & 1) && (board >> n & 1)) {
                            reach |= bit(n);
                            layer |= bit(n);
                        }
                    });
                }
            }
        }
        Status status() const {  // hex/core.rs:314-322
            if (winner) return {true, winner};
            if (empties == 0) return Status::draw();
            return Status::ongoing();
        }
        void legal_moves(std::vector<Move>& out) const {  // ascending idx, hex/core.rs:297-305
            out.clear();
            const u128 occ = red | blue;
            for (int i = 0; i < N * N; i++)
                if (!(occ >> i & 1)) out.push_back(Move{(uint8_t)i});
        }
        Position moved(Move m) const {  // make_move, hex/core.rs:278-291
            Position r = *this;
            (to_move == PLAYER1 ? r.red : r.blue) |= bit(m.idx);
            r.update_reach(m.idx / N, m.idx % N, to_move);
            r.empties--;
            r.to_move = opposite(to_move);
            return r;
        }
        Position flipped() const {  // hex/core.rs:324-334
            Position r;
            r.red = transpose(blue), r.blue = transpose(red);
            r.to_move = opposite(to_move);
            r.left_red_reach = transpose(top_blue_reach), r.top_blue_reach = transpose(left_red_reach);
            r.empties = empties, r.winner = (int8_t)-winner;
            return r;
        }
        void planes(uint64_t* out) const {  // hex/net.rs:14-24 + lo,hi split of serialize/hex.rs:19-24
            const u128 full = (bit(N * N)) - 1;
            const u128 p[3] = {red, blue, full};
            for (int i = 0; i < 3; i++) out[2 * i] = (uint64_t)p[i], out[2 * i + 1] = (uint64_t)(p[i] >> 64);
        }
        bool operator==(const Position& p) const {
            return red == p.red && blue == p.blue && to_move == p.to_move && left_red_reach == p.left_red_reach &&
                   top_blue_reach == p.top_blue_reach && empties == p.empties && winner == p.winner;
        }
        uint64_t hash() const {
            uint64_t h = mix64((uint64_t)red ^ 0x9E3779B97F4A7C15ull);
            h = mix64(h ^ (uint64_t)(red >> 64));
            h = mix64(h ^ (uint64_t)blue);
            h = mix64(h ^ (uint64_t)(blue >> 64));
            return mix64(h ^ to_move);
        }
    };
};

}  // namespace cattus
