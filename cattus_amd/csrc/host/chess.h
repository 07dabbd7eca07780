// Chess rules for the host-side search.
//
// The reference wraps the third-party `chess` crate 3.2.0 (engine/src/chess/core.rs:10-451); that
// crate is not part of the reference tree, so move generation, make-move, status and hashing are
// written here from the rules of chess, and the crate behaviours the reference depends on are
// restated from its published behaviour (parity unpinned where noted):
//   * en_passant() is the square of the PAWN that just moved two squares, and it is recorded only if
//     an enemy pawn stands next to it (Board::set_ep) -- this is what plane 16 and position equality
//     see (chess/net/mod.rs:54, chess/core.rs:261-262,288-305);
//   * legal-move iteration order (it fixes child insertion order in the search): piece kinds in the
//     order pawn, knight, bishop, rook, queen, king; sources ascending; destinations ascending;
//     promotions queen, knight, rook, bishop.  [unpinned]
// What IS pinned by the reference's own tests is restated in tests/test_host_rules.py
// (chess/core.rs:617-729): a mating line, the custom fifty-move counter, flip involution and
// flip/legal-move commutation; plus the public perft node counts.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

#include "games.h"

namespace cattus {

namespace chessimpl {

enum Piece : uint8_t { PAWN = 0, KNIGHT, BISHOP, ROOK, QUEEN, KING, NO_PIECE };
enum : uint8_t { WK = 1, WQ = 2, BK = 4, BQ = 8 };  // castle rights

inline int lsb(uint64_t b) { return __builtin_ctzll(b); }
inline int popcnt(uint64_t b) { return __builtin_popcountll(b); }
inline uint64_t sq_bb(int s) { return 1ull << s; }

struct Tables {
    uint64_t knight[64], king[64], pawn_att[2][64], ray[8][64], between[64][64], line[64][64];
    uint64_t zob_piece[2][6][64], zob_castle[16], zob_ep[64], zob_side;
    // NN move index: 1792 queen/knight-geometry (src,dst) pairs + 88 promotions (chess/core.rs:453-605)
    uint16_t move_to_nn[64 * 64 + 22 * 4];
    uint16_t nn_to_move[1880][3];  // src, dst, promo piece (NO_PIECE if none)
    Tables() {
        static const int dr[8] = {1, 1, 0, -1, -1, -1, 0, 1}, df[8] = {0, 1, 1, 1, 0, -1, -1, -1};  // N NE E SE S SW W NW
        for (int s = 0; s < 64; s++) {
            const int r = s / 8, f = s % 8;
            knight[s] = king[s] = pawn_att[0][s] = pawn_att[1][s] = 0;
            static const int kr[8] = {2, 2, 1, 1, -1, -1, -2, -2}, kf[8] = {1, -1, 2, -2, 2, -2, 1, -1};
            for (int i = 0; i < 8; i++) {
                int rr = r + kr[i], ff = f + kf[i];
                if (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) knight[s] |= sq_bb(rr * 8 + ff);
                rr = r + dr[i], ff = f + df[i];
                if (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) king[s] |= sq_bb(rr * 8 + ff);
            }
            for (int d = -1; d <= 1; d += 2) {
                if (f + d >= 0 && f + d < 8) {
                    if (r + 1 < 8) pawn_att[0][s] |= sq_bb((r + 1) * 8 + f + d);
                    if (r - 1 >= 0) pawn_att[1][s] |= sq_bb((r - 1) * 8 + f + d);
                }
            }
            for (int d = 0; d < 8; d++) {
                uint64_t b = 0;
                int rr = r + dr[d], ff = f + df[d];
                while (rr >= 0 && rr < 8 && ff >= 0 && ff < 8) {
                    b |= sq_bb(rr * 8 + ff);
                    rr += dr[d], ff += df[d];
                }
                ray[d][s] = b;
            }
        }
        for (int a = 0; a < 64; a++)
            for (int b = 0; b < 64; b++) {
                between[a][b] = line[a][b] = 0;
                for (int d = 0; d < 8; d++)
                    if (ray[d][a] & sq_bb(b)) {
                        between[a][b] = ray[d][a] & ray[(d + 4) & 7][b];
                        line[a][b] = ray[d][a] | ray[(d + 4) & 7][a] | sq_bb(a);
                    }
            }
        uint64_t z = 0x7A3C5D1E9B2F4861ull;
        auto nextz = [&]() {
            z += 0x9E3779B97F4A7C15ull;
            return mix64(z);
        };
        for (auto& c : zob_piece)
            for (auto& p : c)
                for (auto& s : p) s = nextz();
        for (auto& c : zob_castle) c = nextz();
        for (auto& e : zob_ep) e = nextz();
        zob_side = nextz();

        // NN index table, generated instead of listed: for every source square in index order, all
        // queen- and knight-reachable destinations in index order; then, per source file a..h and
        // destination file df = sf-1, sf, sf+1, the promotions q, r, b, n from rank 7 to rank 8.
        for (auto& m : move_to_nn) m = 0xffff;
        int idx = 0;
        for (int s = 0; s < 64; s++) {
            uint64_t dests = knight[s];
            for (int d = 0; d < 8; d++) dests |= ray[d][s];
            for (int t = 0; t < 64; t++)
                if (dests & sq_bb(t)) {
                    move_to_nn[s * 64 + t] = (uint16_t)idx;
                    nn_to_move[idx][0] = (uint16_t)s, nn_to_move[idx][1] = (uint16_t)t, nn_to_move[idx][2] = NO_PIECE;
                    idx++;
                }
        }
        static const Piece promo_order[4] = {QUEEN, ROOK, BISHOP, KNIGHT};
        for (int sf = 0; sf < 8; sf++)
            for (int dfl = sf - 1; dfl <= sf + 1; dfl++) {
                if (dfl < 0 || dfl > 7) continue;
                for (int k = 0; k < 4; k++) {
                    move_to_nn[64 * 64 + (sf * 2 + dfl) * 4 + k] = (uint16_t)idx;  // to_idx, chess/core.rs:55-72
                    nn_to_move[idx][0] = (uint16_t)(48 + sf), nn_to_move[idx][1] = (uint16_t)(56 + dfl);
                    nn_to_move[idx][2] = promo_order[k];
                    idx++;
                }
            }
    }
};

inline const Tables& tables() {
    static const Tables t;
    return t;
}

inline uint64_t rook_attacks(int s, uint64_t occ) {
    const Tables& T = tables();
    uint64_t a = 0, r, b;
    r = T.ray[0][s]; a |= r; b = r & occ; if (b) a &= ~T.ray[0][lsb(b)];
    r = T.ray[2][s]; a |= r; b = r & occ; if (b) a &= ~T.ray[2][lsb(b)];
    r = T.ray[4][s]; b = r & occ; a |= b ? r & ~T.ray[4][63 - __builtin_clzll(b)] : r;
    r = T.ray[6][s]; b = r & occ; a |= b ? r & ~T.ray[6][63 - __builtin_clzll(b)] : r;
    return a;
}
inline uint64_t bishop_attacks(int s, uint64_t occ) {
    const Tables& T = tables();
    uint64_t a = 0, r, b;
    r = T.ray[1][s]; a |= r; b = r & occ; if (b) a &= ~T.ray[1][lsb(b)];
    r = T.ray[7][s]; a |= r; b = r & occ; if (b) a &= ~T.ray[7][lsb(b)];
    r = T.ray[3][s]; b = r & occ; a |= b ? r & ~T.ray[3][63 - __builtin_clzll(b)] : r;
    r = T.ray[5][s]; b = r & occ; a |= b ? r & ~T.ray[5][63 - __builtin_clzll(b)] : r;
    return a;
}

}  // namespace chessimpl

struct ChessGame {
    static constexpr int BOARD = 8, MOVES = 1880, PLANES = 18, PLANE_WORDS = 1, REPETITION_LIMIT = 3;
    static const char* name() { return "chess"; }
    typedef chessimpl::Piece Piece;

    struct Move {
        uint8_t src, dst, promo;  // promo = NO_PIECE if none
        Move flipped() const { return Move{(uint8_t)(src ^ 56), (uint8_t)(dst ^ 56), promo}; }  // chess/core.rs:82-92
        int to_idx() const {  // chess/core.rs:55-72
            if (promo != chessimpl::NO_PIECE) {
                const int off = promo == chessimpl::QUEEN ? 0 : promo == chessimpl::ROOK ? 1 : promo == chessimpl::BISHOP ? 2 : 3;
                return 64 * 64 + ((src & 7) * 2 + (dst & 7)) * 4 + off;
            }
            return src * 64 + dst;
        }
        int nn_idx() const { return chessimpl::tables().move_to_nn[to_idx()]; }
        bool operator==(const Move& o) const { return src == o.src && dst == o.dst && promo == o.promo; }
        std::string str() const {
            std::string s;
            s += (char)('a' + (src & 7)), s += (char)('1' + (src >> 3)), s += (char)('a' + (dst & 7)), s += (char)('1' + (dst >> 3));
            if (promo != chessimpl::NO_PIECE) s += "pnbrqk"[promo];
            return s;
        }
    };

    struct Position {
        uint64_t pieces[6] = {0, 0, 0, 0, 0, 0};
        uint64_t color[2] = {0, 0};
        uint64_t key = 0;     // zobrist of everything operator== compares
        uint8_t stm = 0;      // 0 white (Player1), 1 black
        uint8_t castle = 0;
        int8_t ep = -1;       // square of the pawn that can be captured en passant, or -1
        uint8_t fifty = 0;    // the reference's own counter (chess/core.rs:336-343)

        static Position initial() { return from_fen("rnbqkbnr/pppppppp/8/8/8/8/PPPPPPPP/RNBQKBNR w KQkq - 0 1"); }
        Color turn() const { return stm == 0 ? PLAYER1 : PLAYER2; }
        uint64_t occ() const { return color[0] | color[1]; }

        void put(int c, int p, int s) {
            pieces[p] |= 1ull << s;
            color[c] |= 1ull << s;
        }
        void rehash() {
            const auto& T = chessimpl::tables();
            key = 0;
            for (int c = 0; c < 2; c++)
                for (int p = 0; p < 6; p++) {
                    uint64_t b = pieces[p] & color[c];
                    while (b) {
                        key ^= T.zob_piece[c][p][chessimpl::lsb(b)];
                        b &= b - 1;
                    }
                }
            key ^= T.zob_castle[castle];
            if (ep >= 0) key ^= T.zob_ep[ep];
            if (stm) key ^= T.zob_side;
        }
        // Board::set_ep: keep the pawn square only if a pawn of the side to move can capture it
        void set_ep(int pawn_sq) {
            const int f = pawn_sq & 7;
            uint64_t adj = 0;
            if (f > 0) adj |= 1ull << (pawn_sq - 1);
            if (f < 7) adj |= 1ull << (pawn_sq + 1);
            ep = (adj & pieces[chessimpl::PAWN] & color[stm]) ? (int8_t)pawn_sq : (int8_t)-1;
        }

        static Position from_fen(const std::string& fen) {
            Position p;
            size_t i = 0;
            int r = 7, f = 0;
            for (; i < fen.size() && fen[i] != ' '; i++) {
                const char ch = fen[i];
                if (ch == '/') {
                    r--, f = 0;
                } else if (ch >= '1' && ch <= '8') {
                    f += ch - '0';
                } else {
                    const char* names = "pnbrqk";
                    const char lo = (char)(ch | 0x20);
                    int pc = 0;
                    while (pc < 6 && names[pc] != lo) pc++;
                    if (pc < 6 && r >= 0 && f < 8) p.put(ch >= 'a' ? 1 : 0, pc, r * 8 + f);
                    f++;
                }
            }
            while (i < fen.size() && fen[i] == ' ') i++;
            p.stm = (i < fen.size() && fen[i] == 'b') ? 1 : 0;
            while (i < fen.size() && fen[i] != ' ') i++;
            while (i < fen.size() && fen[i] == ' ') i++;
            for (; i < fen.size() && fen[i] != ' '; i++) {
                if (fen[i] == 'K') p.castle |= chessimpl::WK;
                if (fen[i] == 'Q') p.castle |= chessimpl::WQ;
                if (fen[i] == 'k') p.castle |= chessimpl::BK;
                if (fen[i] == 'q') p.castle |= chessimpl::BQ;
            }
            while (i < fen.size() && fen[i] == ' ') i++;
            if (i < fen.size() && fen[i] != '-') {
                // only the file is read; the pawn stands on the 4th rank of the side that just moved
                const int file = fen[i] - 'a';
                p.set_ep((p.stm == 0 ? 4 : 3) * 8 + file);
            }
            p.rehash();
            return p;
        }

        int piece_on(int s) const {
            const uint64_t b = 1ull << s;
            for (int p = 0; p < 6; p++)
                if (pieces[p] & b) return p;
            return chessimpl::NO_PIECE;
        }
        uint64_t attackers_to(int s, uint64_t occupied, int by) const {
            using namespace chessimpl;
            const auto& T = tables();
            const uint64_t them = color[by];
            uint64_t a = T.pawn_att[by ^ 1][s] & pieces[PAWN];
            a |= T.knight[s] & pieces[KNIGHT];
            a |= T.king[s] & pieces[KING];
            a |= rook_attacks(s, occupied) & (pieces[ROOK] | pieces[QUEEN]);
            a |= bishop_attacks(s, occupied) & (pieces[BISHOP] | pieces[QUEEN]);
            return a & them;
        }
        bool in_check() const { return attackers_to(chessimpl::lsb(pieces[chessimpl::KING] & color[stm]), occ(), stm ^ 1) != 0; }

        // Legal moves in the iteration order documented at the top of this file.
        void legal_moves(std::vector<Move>& out) const {
            using namespace chessimpl;
            out.clear();
            const auto& T = tables();
            const int us = stm, them = stm ^ 1;
            const uint64_t own = color[us], enemy = color[them], all = own | enemy;
            const int ksq = lsb(pieces[KING] & own);
            const uint64_t checkers = attackers_to(ksq, all, them);
            // pinned pieces: own piece alone between king and an enemy slider on a shared line
            uint64_t pinned = 0;
            uint64_t snipers = ((T.ray[0][ksq] | T.ray[2][ksq] | T.ray[4][ksq] | T.ray[6][ksq]) & (pieces[ROOK] | pieces[QUEEN]) & enemy) |
                               ((T.ray[1][ksq] | T.ray[3][ksq] | T.ray[5][ksq] | T.ray[7][ksq]) & (pieces[BISHOP] | pieces[QUEEN]) & enemy);
            while (snipers) {
                const int s = lsb(snipers);
                snipers &= snipers - 1;
                const uint64_t b = T.between[ksq][s] & all;
                if (b && !(b & (b - 1)) && (b & own)) pinned |= b;
            }
            // squares a non-king move may go to: anywhere, or capture/block a single checker
            uint64_t target = ~own;
            if (checkers) {
                if (checkers & (checkers - 1)) target = 0;  // double check: king moves only
                else target &= checkers | T.between[ksq][lsb(checkers)];
            }
            auto emit = [&](int s, uint64_t dests, bool promo) {
                while (dests) {
                    const int d = lsb(dests);
                    dests &= dests - 1;
                    if (promo) {
                        out.push_back(Move{(uint8_t)s, (uint8_t)d, QUEEN});
                        out.push_back(Move{(uint8_t)s, (uint8_t)d, KNIGHT});
                        out.push_back(Move{(uint8_t)s, (uint8_t)d, ROOK});
                        out.push_back(Move{(uint8_t)s, (uint8_t)d, BISHOP});
                    } else {
                        out.push_back(Move{(uint8_t)s, (uint8_t)d, NO_PIECE});
                    }
                }
            };
            // pawns
            {
                uint64_t pawns = pieces[PAWN] & own;
                const int up = us == 0 ? 8 : -8;
                const int start_rank = us == 0 ? 1 : 6, promo_rank = us == 0 ? 6 : 1;
                while (pawns) {
                    const int s = lsb(pawns);
                    pawns &= pawns - 1;
                    uint64_t dests = 0;
                    const int one = s + up;
                    if (!(all >> one & 1)) {
                        dests |= 1ull << one;
                        if ((s >> 3) == start_rank && !(all >> (one + up) & 1)) dests |= 1ull << (one + up);
                    }
                    dests |= T.pawn_att[us][s] & enemy;
                    dests &= target;
                    if (pinned >> s & 1) dests &= T.line[ksq][s];
                    // en passant: destination is the square behind the captured pawn
                    if (ep >= 0 && (T.pawn_att[us][s] >> (ep + up) & 1)) {
                        const int d = ep + up;
                        const uint64_t occ2 = (all ^ (1ull << s) ^ (1ull << ep)) | (1ull << d);
                        // legal iff the king is not attacked afterwards (the captured pawn is gone)
                        Position tmp = *this;
                        tmp.pieces[PAWN] &= ~(1ull << ep);
                        tmp.color[them] &= ~(1ull << ep);
                        if (!tmp.attackers_to(ksq, occ2, them)) dests |= 1ull << d;
                    }
                    emit(s, dests, (s >> 3) == promo_rank);
                }
            }
            // knights, bishops, rooks, queens
            for (int pc = KNIGHT; pc <= QUEEN; pc++) {
                uint64_t bb = pieces[pc] & own;
                while (bb) {
                    const int s = lsb(bb);
                    bb &= bb - 1;
                    uint64_t dests = pc == KNIGHT ? T.knight[s]
                                     : pc == BISHOP ? bishop_attacks(s, all)
                                     : pc == ROOK   ? rook_attacks(s, all)
                                                    : (rook_attacks(s, all) | bishop_attacks(s, all));
                    dests &= target;
                    if (pinned >> s & 1) dests &= T.line[ksq][s];
                    emit(s, dests, false);
                }
            }
            // king (castling destinations are part of the king's destination set)
            {
                uint64_t dests = T.king[ksq] & ~own, legal = 0;
                const uint64_t occ_nok = all ^ (1ull << ksq);
                while (dests) {
                    const int d = lsb(dests);
                    dests &= dests - 1;
                    if (!attackers_to(d, occ_nok, them)) legal |= 1ull << d;
                }
                if (!checkers) {
                    const int base = us == 0 ? 0 : 56;
                    const uint8_t ks = us == 0 ? WK : BK, qs = us == 0 ? WQ : BQ;
                    if ((castle & ks) && !(all & (3ull << (base + 5))) && (legal >> (base + 5) & 1) &&
                        !attackers_to(base + 6, all, them))
                        legal |= 1ull << (base + 6);
                    if ((castle & qs) && !(all & (7ull << (base + 1))) && (legal >> (base + 3) & 1) &&
                        !attackers_to(base + 2, all, them))
                        legal |= 1ull << (base + 2);
                }
                emit(ksq, legal, false);
            }
        }

        // make_move_new + the reference's fifty-move bookkeeping (chess/core.rs:327-346)
        Position moved(Move m) const {
            using namespace chessimpl;
            const auto& T = tables();
            Position n = *this;
            const int us = stm, them = stm ^ 1;
            const int pc = piece_on(m.src), cap = piece_on(m.dst);
            const uint64_t from = 1ull << m.src, to = 1ull << m.dst;
            n.key ^= T.zob_castle[castle];
            if (ep >= 0) n.key ^= T.zob_ep[ep];
            if (cap != NO_PIECE) {
                n.pieces[cap] &= ~to, n.color[them] &= ~to;
                n.key ^= T.zob_piece[them][cap][m.dst];
            }
            n.pieces[pc] &= ~from, n.color[us] &= ~from;
            n.key ^= T.zob_piece[us][pc][m.src];
            const int placed = m.promo != NO_PIECE ? m.promo : pc;
            n.pieces[placed] |= to, n.color[us] |= to;
            n.key ^= T.zob_piece[us][placed][m.dst];
            if (pc == PAWN && ep >= 0 && m.dst == ep + (us == 0 ? 8 : -8) && cap == NO_PIECE && (m.src & 7) != (m.dst & 7)) {
                n.pieces[PAWN] &= ~(1ull << ep), n.color[them] &= ~(1ull << ep);  // en-passant capture
                n.key ^= T.zob_piece[them][PAWN][ep];
            }
            if (pc == KING && (m.dst - m.src == 2 || m.src - m.dst == 2)) {  // castling: move the rook
                const int rf = m.dst > m.src ? m.src + 3 : m.src - 4, rt = m.dst > m.src ? m.src + 1 : m.src - 1;
                n.pieces[ROOK] ^= (1ull << rf) | (1ull << rt), n.color[us] ^= (1ull << rf) | (1ull << rt);
                n.key ^= T.zob_piece[us][ROOK][rf] ^ T.zob_piece[us][ROOK][rt];
            }
            // castle rights: lost when the king or a rook leaves its square or a rook is captured there
            auto touch = [&](int s) {
                if (s == 4) n.castle &= ~(WK | WQ);
                if (s == 60) n.castle &= ~(BK | BQ);
                if (s == 7) n.castle &= ~WK;
                if (s == 0) n.castle &= ~WQ;
                if (s == 63) n.castle &= ~BK;
                if (s == 56) n.castle &= ~BQ;
            };
            touch(m.src), touch(m.dst);
            n.key ^= T.zob_castle[n.castle];
            n.stm = (uint8_t)them;
            n.key ^= T.zob_side;
            n.ep = -1;
            if (pc == PAWN && (m.dst - m.src == 16 || m.src - m.dst == 16)) n.set_ep(m.dst);
            if (n.ep >= 0) n.key ^= T.zob_ep[n.ep];
            const bool is_pawn = pc == PAWN, is_atk = cap != NO_PIECE;
            n.fifty = (is_pawn || is_atk) ? 0 : (us == 0 ? (uint8_t)(fifty + 1) : fifty);
            return n;
        }

        Status status() const {  // chess/core.rs:348-364
            std::vector<Move> mv;
            legal_moves(mv);
            if (!mv.empty() && fifty < 50) return Status::ongoing();
            if (mv.empty()) return in_check() ? Status::won(opposite(turn())) : Status::draw();
            return Status::draw();
        }

        Position flipped() const {  // chess/core.rs:366-399
            Position n;
            for (int p = 0; p < 6; p++) n.pieces[p] = __builtin_bswap64(pieces[p]);
            n.color[0] = __builtin_bswap64(color[1]);
            n.color[1] = __builtin_bswap64(color[0]);
            n.stm = stm ^ 1;
            n.castle = (uint8_t)(((castle & 3) << 2) | ((castle >> 2) & 3));
            n.fifty = fifty;
            n.ep = -1;
            if (ep >= 0) n.set_ep((n.stm == 0 ? 4 : 3) * 8 + (ep & 7));  // BoardBuilder keeps the file only
            n.rehash();
            return n;
        }

        void planes(uint64_t* out) const {  // chess/net/mod.rs:19-60
            for (int p = 0; p < 6; p++) out[p] = pieces[p] & color[0], out[6 + p] = pieces[p] & color[1];
            out[12] = castle & chessimpl::WK ? ~0ull : 0;
            out[13] = castle & chessimpl::WQ ? ~0ull : 0;
            out[14] = castle & chessimpl::BK ? ~0ull : 0;
            out[15] = castle & chessimpl::BQ ? ~0ull : 0;
            out[16] = ep >= 0 ? 1ull << ep : 0;
            out[17] = ~0ull;
        }
        // PartialEq of the reference ignores the fifty-move counter (chess/core.rs:288-305)
        bool operator==(const Position& o) const {
            if (key != o.key) return false;
            for (int p = 0; p < 6; p++)
                if (pieces[p] != o.pieces[p]) return false;
            return color[0] == o.color[0] && color[1] == o.color[1] && castle == o.castle && ep == o.ep && stm == o.stm;
        }
        uint64_t hash() const { return key; }

        std::string fen() const {  // chess/core.rs:179-277 (no move counters)
            std::string s;
            for (int r = 7; r >= 0; r--) {
                int blanks = 0;
                for (int f = 0; f < 8; f++) {
                    const int sq = r * 8 + f, pc = piece_on(sq);
                    if (pc == chessimpl::NO_PIECE) { blanks++; continue; }
                    if (blanks) s += (char)('0' + blanks), blanks = 0;
                    const char c = "pnbrqk"[pc];
                    s += (color[0] >> sq & 1) ? (char)(c - 32) : c;
                }
                if (blanks) s += (char)('0' + blanks);
                if (r) s += '/';
            }
            s += stm ? " b " : " w ";
            std::string cr;
            if (castle & chessimpl::WK) cr += 'K';
            if (castle & chessimpl::WQ) cr += 'Q';
            if (castle & chessimpl::BK) cr += 'k';
            if (castle & chessimpl::BQ) cr += 'q';
            s += cr.empty() ? "-" : cr;
            s += ' ';
            if (ep >= 0) {
                s += (char)('a' + (ep & 7));
                s += (char)('1' + (ep >> 3) + (stm == 0 ? 1 : -1));
            } else {
                s += '-';
            }
            return s;
        }
    };

    static uint64_t perft(const Position& p, int depth) {
        std::vector<Move> mv;
        p.legal_moves(mv);
        if (depth <= 1) return depth == 1 ? mv.size() : 1;
        uint64_t n = 0;
        for (auto& m : mv) n += perft(p.moved(m), depth - 1);
        return n;
    }
};

}  // namespace cattus
