// C ABI of libcattus_selfplay.so (include/cattus_selfplay.h).
#include <algorithm>
#include <memory>
#include <string>

#include "../../../include/cattus_selfplay.h"
#include "chess.h"
#include "games.h"
#include "selfplay.h"

using namespace cattus;

#define SP_API extern "C" __attribute__((visibility("default")))

namespace {

thread_local std::string g_err;

template <typename F>
auto dispatch(int game, F&& f) -> decltype(f(TttGame{})) {
    switch (game) {
        case CATTUS_GAME_TTT: return f(TttGame{});
        case CATTUS_GAME_HEX4: return f(HexGame<4>{});
        case CATTUS_GAME_HEX5: return f(HexGame<5>{});
        case CATTUS_GAME_HEX7: return f(HexGame<7>{});
        case CATTUS_GAME_HEX9: return f(HexGame<9>{});
        case CATTUS_GAME_HEX11: return f(HexGame<11>{});
        case CATTUS_GAME_CHESS: return f(ChessGame{});
        default: g_err = "unknown game id"; return decltype(f(TttGame{}))();
    }
}

SelfPlayConfig to_config(const cattus_sp_config* c, uint32_t games_num = 0) {
    SelfPlayConfig cfg;
    cfg.mcts.sim_num = c->sim_num;
    cfg.mcts.explore_factor = c->explore_factor;
    // temperature_policy: all entries but the last are (threshold, T); the last entry's T is the tail
    // (self_play_cmd.rs:73-76)
    const uint32_t n = c->temperature_count;  // 1..8, checked by config_error()
    for (uint32_t i = 0; i + 1 < n; i++) cfg.mcts.temperature.scheduled.emplace_back(c->temperature_threshold[i], c->temperature_value[i]);
    cfg.mcts.temperature.last = n ? c->temperature_value[n - 1] : 0.0f;
    cfg.mcts.prior_noise_alpha = c->prior_noise_alpha;
    cfg.mcts.prior_noise_epsilon = c->prior_noise_epsilon;
    cfg.cache_size = c->cache_size;
    cfg.batch_size = std::max(1u, c->batch_size);
    cfg.threads = std::max(1u, c->threads);
    cfg.concurrent_games = c->concurrent_games;
    cfg.seed = c->seed;
    cfg.first_game = c->first_game;
    cfg.game_stride = std::max(1u, c->game_stride);
    cfg.host_alloc = c->host_alloc;
    cfg.host_free = c->host_free;
    cfg.eval_threads = c->eval_threads ? std::min(c->eval_threads, 256u) : 2;
    cfg.mcts.leaves_in_flight = std::max(1u, std::min(c->leaves_in_flight, 16u));
    cfg.max_game_plies = c->max_game_plies;
    if (c->game_list) cfg.game_list.assign(c->game_list, c->game_list + games_num);
    if (c->progress_path) cfg.progress_path = c->progress_path;
    return cfg;
}

// What TemperaturePolicy::scheduled asserts (mcts/mod.rs:463-476) plus the limits of the C struct; nullptr = ok.
const char* config_error(const cattus_sp_config* c) {
    if (!c || c->struct_size != sizeof(cattus_sp_config)) return "bad config struct";
    if (c->sim_num < 2) return "sim_num must be > 1";
    if (c->temperature_count < 1 || c->temperature_count > 8) return "temperature_count must be 1..8";
    for (uint32_t i = 0; i < c->temperature_count; i++) {
        if (!(c->temperature_value[i] >= 0.0f)) return "temperatures must be >= 0";
        if (i > 0 && i + 1 < c->temperature_count && c->temperature_threshold[i] <= c->temperature_threshold[i - 1])
            return "temperature thresholds must be strictly increasing";
    }
    if (!(c->explore_factor >= 0.0f) || !(c->prior_noise_alpha >= 0.0f)) return "explore_factor / prior_noise_alpha must be >= 0";
    if (!(c->prior_noise_epsilon >= 0.0f && c->prior_noise_epsilon <= 1.0f)) return "prior_noise_epsilon must be in [0, 1]";
    return nullptr;
}

// ---- position handles ----
struct PosBase {
    virtual ~PosBase() {}
    virtual int status() const = 0;
    virtual int turn() const = 0;
    virtual int legal(uint16_t* nn, char* names, int cap) const = 0;
    virtual PosBase* moved(int k) const = 0;
    virtual PosBase* flipped() const = 0;
    virtual bool equal(const PosBase* o) const = 0;
    virtual int planes(uint64_t* out) const = 0;
    virtual std::string str() const = 0;
    virtual int flipped_move_nn(int k) const = 0;
    virtual int test_record(uint8_t* out, int cap) const = 0;
    int game = 0;
};

template <typename G>
std::string pos_string(const typename G::Position& p);
template <>
std::string pos_string<TttGame>(const TttGame::Position& p) {
    std::string s;
    for (int i = 0; i < 9; i++) s += (p.x >> i & 1) ? 'x' : (p.o >> i & 1) ? 'o' : '_';
    s += p.to_move == PLAYER1 ? 'x' : 'o';
    return s;
}
template <>
std::string pos_string<ChessGame>(const ChessGame::Position& p) {
    return p.fen();
}
template <typename G>
std::string pos_string(const typename G::Position& p) {
    std::string s;
    for (int i = 0; i < G::MOVES; i++) s += (p.red >> i & 1) ? 'r' : (p.blue >> i & 1) ? 'b' : 'e';
    s += p.to_move == PLAYER1 ? 'r' : 'b';
    return s;
}

template <typename G>
struct PosImpl : PosBase {
    typename G::Position p;
    int status() const override {
        const Status s = p.status();
        return s.finished ? s.winner : 2;
    }
    int turn() const override { return p.turn(); }
    int legal(uint16_t* nn, char* names, int cap) const override {
        std::vector<typename G::Move> mv;
        p.legal_moves(mv);
        for (int i = 0; i < (int)mv.size() && i < cap; i++) {
            if (nn) nn[i] = (uint16_t)mv[i].nn_idx();
            if (names) {
                const std::string s = mv[i].str();
                memset(names + 12 * i, 0, 12);
                memcpy(names + 12 * i, s.data(), std::min<size_t>(11, s.size()));
            }
        }
        return (int)mv.size();
    }
    PosBase* moved(int k) const override {
        std::vector<typename G::Move> mv;
        p.legal_moves(mv);
        if (k < 0 || k >= (int)mv.size()) return nullptr;
        auto* r = new PosImpl<G>;
        r->game = game, r->p = p.moved(mv[k]);
        return r;
    }
    PosBase* flipped() const override {
        auto* r = new PosImpl<G>;
        r->game = game, r->p = p.flipped();
        return r;
    }
    bool equal(const PosBase* o) const override { return o->game == game && static_cast<const PosImpl<G>*>(o)->p == p; }
    int planes(uint64_t* out) const override {
        p.planes(out);
        return G::PLANES * G::PLANE_WORDS;
    }
    std::string str() const override { return pos_string<G>(p); }
    int flipped_move_nn(int k) const override {
        std::vector<typename G::Move> mv;
        p.legal_moves(mv);
        if (k < 0 || k >= (int)mv.size()) return -1;
        return mv[k].flipped().nn_idx();
    }
    int test_record(uint8_t* out, int cap) const override {
        if (cap < (int)Serializer<G>::RECORD_BYTES || p.turn() != PLAYER1) return -1;
        std::vector<typename G::Move> mv;
        p.legal_moves(mv);
        const size_t n = mv.size();
        std::vector<std::pair<typename G::Move, float>> probs;
        for (size_t i = 0; i < n; i++) probs.emplace_back(mv[i], (float)i / (float)(n * (n - 1)));
        const int8_t winner = n % 3 == 0 ? 1 : n % 3 == 1 ? -1 : 0;
        Serializer<G>::serialize(p, probs, winner, out);
        return (int)Serializer<G>::RECORD_BYTES;
    }
};

template <typename G>
bool parse_pos(const char* str, typename G::Position& out);
template <>
bool parse_pos<TttGame>(const char* str, TttGame::Position& out) {  // test_util.rs:7-34
    if (strlen(str) != 10) return false;
    out = TttGame::Position();
    for (int i = 0; i < 9; i++) {
        if (str[i] == 'x') out.x |= (uint16_t)(1u << i);
        else if (str[i] == 'o') out.o |= (uint16_t)(1u << i);
        else if (str[i] != '_') return false;
    }
    out.to_move = str[9] == 'x' ? PLAYER1 : PLAYER2;
    out.check_winner();
    return str[9] == 'x' || str[9] == 'o';
}
template <>
bool parse_pos<ChessGame>(const char* str, ChessGame::Position& out) {
    out = ChessGame::Position::from_fen(str);
    return true;
}
template <typename G>
bool parse_pos(const char* str, typename G::Position& out) {  // test_util.rs:36-66
    if ((int)strlen(str) != G::MOVES + 1) return false;
    u128 red = 0, blue = 0;
    for (int i = 0; i < G::MOVES; i++) {
        if (str[i] == 'r') red |= (u128)1 << i;
        else if (str[i] == 'b') blue |= (u128)1 << i;
        else if (str[i] != 'e') return false;
    }
    const char t = str[G::MOVES];
    if (t != 'r' && t != 'b') return false;
    out = G::Position::from_board(red, blue, t == 'r' ? PLAYER1 : PLAYER2);
    return true;
}

}  // namespace

struct cattus_pos {
    std::unique_ptr<PosBase> p;
};

struct cattus_sp_result {
    cattus_sp_summary summary{};
    std::vector<Record> records;
    size_t record_bytes = 0;
};

SP_API const char* cattus_sp_last_error(void) { return g_err.c_str(); }

SP_API int cattus_sp_game_info(int game, uint32_t out[5]) {
    return dispatch(game, [&](auto g) -> int {
        typedef decltype(g) G;
        out[0] = G::BOARD, out[1] = G::MOVES, out[2] = G::PLANES, out[3] = G::PLANE_WORDS;
        out[4] = (uint32_t)Serializer<G>::RECORD_BYTES;
        return 1;
    }) ? 0 : -1;
}

SP_API int cattus_sp_run(int game, const cattus_sp_config* c, cattus_net_eval_fn net1, void* ctx1, cattus_net_eval_fn net2,
                         void* ctx2, uint32_t games_num, const char* out_dir1, const char* out_dir2, int keep_records,
                         cattus_sp_result** out) {
    if (!net1 || !out) {
        g_err = "bad arguments";
        return -1;
    }
    if (const char* why = config_error(c)) {
        g_err = why;
        return -1;
    }
    *out = nullptr;
    int rc = -1;
    dispatch(game, [&](auto g) -> int {
        typedef decltype(g) G;
        const bool same = net2 == nullptr;
        const NetHandle h1{net1, ctx1, c->legal_net1};
        const NetHandle h2 = same ? h1 : NetHandle{net2, ctx2, c->legal_net2};
        SelfPlayRunner<G> runner(to_config(c, games_num), h1, h2, same);
        auto res = std::make_unique<cattus_sp_result>();
        SelfPlayResult r;
        rc = runner.generate_data(games_num, out_dir1 ? out_dir1 : "", out_dir2 ? out_dir2 : "", &res->records, r);
        if (rc != 0) {
            g_err = runner.error();
            return 1;
        }
        Metrics& m = runner.metrics();
        cattus_sp_summary& s = res->summary;
        s.player1_wins = r.w1, s.player2_wins = r.w2, s.draws = r.d;
        s.positions = r.positions, s.records = res->records.size();
        s.activation_count = m.activation_count, s.node_evals = m.node_evals;
        s.cache_hits = m.cache_hits, s.cache_misses = m.cache_misses;
        s.run_duration = m.run_duration_ema, s.search_duration = m.search_duration_ema;
        s.seconds = r.seconds;
        s.steady_seconds = r.steady_seconds, s.steady_node_evals = r.steady_node_evals;
        s.steady_plies = r.steady_plies, s.steady_batches = r.steady_batches;
        s.adjudicated = r.adjudicated;
        res->record_bytes = Serializer<G>::RECORD_BYTES;
        if (!keep_records) res->records.clear();
        std::sort(res->records.begin(), res->records.end(), [](const Record& a, const Record& b) {
            return a.game_idx != b.game_idx ? a.game_idx < b.game_idx : a.pos_idx < b.pos_idx;
        });
        *out = res.release();
        return 1;
    });
    return rc;
}

SP_API int cattus_sp_result_summary(const cattus_sp_result* r, cattus_sp_summary* out) {
    if (!r || !out) return -1;
    *out = r->summary;
    return 0;
}

SP_API int cattus_sp_result_records(const cattus_sp_result* r, uint8_t* bytes, uint32_t* meta) {
    if (!r) return -1;
    for (size_t i = 0; i < r->records.size(); i++) {
        if (bytes) memcpy(bytes + i * r->record_bytes, r->records[i].bytes.data(), r->record_bytes);
        if (meta) meta[3 * i] = r->records[i].game_idx, meta[3 * i + 1] = r->records[i].pos_idx, meta[3 * i + 2] = r->records[i].dir;
    }
    return (int)r->records.size();
}

SP_API void cattus_sp_result_free(cattus_sp_result* r) { delete r; }

SP_API int cattus_sp_stub_net(void* ctx, const uint64_t* planes, uint32_t n, float* policy, float* value) {
    const uint32_t* c = (const uint32_t*)ctx;
    const uint32_t moves = c[0], words = c[1];
    for (uint32_t b = 0; b < n; b++) {
        uint64_t h = 0x243F6A8885A308D3ull;
        for (uint32_t i = 0; i < words; i++) h = mix64(h ^ planes[(size_t)b * words + i]);
        for (uint32_t m = 0; m < moves; m++) {
            const uint64_t x = mix64(h ^ ((uint64_t)(m + 1) * 0x9E3779B97F4A7C15ull));
            policy[(size_t)b * moves + m] = (float)(x >> 40) * (1.0f / 16777216.0f) * 4.0f - 2.0f;
        }
        const uint64_t x = mix64(h ^ 0xABCDEF0123456789ull);
        value[b] = (float)(x >> 40) * (1.0f / 16777216.0f) * 2.0f - 1.0f;
    }
    return 0;
}

SP_API int64_t cattus_sp_trace_game_ex(int game, const cattus_sp_config* c, cattus_net_eval_fn net, void* ctx, uint32_t max_plies,
                                       const uint16_t* forced, uint32_t n_forced, uint32_t search_from, uint32_t* out, size_t cap) {
    if (!net || !out || (n_forced && !forced)) {
        g_err = "bad arguments";
        return -1;
    }
    if (const char* why = config_error(c)) {
        g_err = why;
        return -1;
    }
    int64_t written = -1;
    dispatch(game, [&](auto g) -> int {
        typedef decltype(g) G;
        SelfPlayConfig cfg = to_config(c);
        cfg.mcts.leaves_in_flight = 1;  // the trace is the sequential search, one evaluation at a time
        Metrics metrics;
        NetValueFunction<G> vf(NetHandle{net, ctx}, cfg.cache_size, &metrics);
        // exactly what one reference worker does for game 0: two persistent players (self_play.rs:180-217)
        MctsPlayer<G> p1(cfg.mcts, cfg.seed * 2 + 1), p2(cfg.mcts, cfg.seed * 2 + 2);
        GameState<G> gs;
        gs.reset(G::Position::initial());
        size_t w = 1;
        uint32_t plies = 0, searched = 0;
        std::vector<float> policy(G::MOVES);
        std::vector<typename G::Move> legal;
        while (searched < max_plies) {
            if (gs.status().finished) break;
            typename G::Move m{};
            bool have_move = false;
            if (plies < n_forced) {
                gs.history.back().legal_moves(legal);
                for (auto& lm : legal)
                    if ((uint16_t)lm.nn_idx() == forced[plies]) m = lm, have_move = true;
                if (!have_move) {
                    g_err = "forced move " + std::to_string(forced[plies]) + " is not legal at ply " + std::to_string(plies);
                    return 0;
                }
            }
            if (plies >= search_from) {
                MctsPlayer<G>& cur = gs.history.back().turn() == PLAYER1 ? p1 : p2;
                cur.begin_search(gs.history);
                while (cur.advance(gs.history) == MctsPlayer<G>::NEED_EVAL) {
                    PendingLeaf<G> pend;
                    Evaluation<G> ev;
                    if (!vf.prepare(cur.pending_position(), pend, ev)) {
                        float value = 0;
                        if (net(ctx, pend.planes, 1, policy.data(), &value) != 0) {
                            g_err = "network callback failed";
                            return 0;
                        }
                        vf.finish(pend, policy.data(), value, ev);
                    }
                    cur.deliver(ev);
                }
                const auto visits = cur.root_visits();
                const auto probs = cur.result();
                typename G::Move chosen;
                if (!cur.choose_move(gs.history, probs, chosen)) {
                    g_err = "search returned no move";
                    return 0;
                }
                if (w + 2 + 2 * visits.size() > cap) {
                    g_err = "trace buffer too small";
                    return 0;
                }
                out[w++] = (uint32_t)chosen.nn_idx();
                out[w++] = (uint32_t)visits.size();
                for (auto& v : visits) out[w++] = (uint32_t)v.first.nn_idx(), out[w++] = v.second;
                searched++;
                if (!have_move) m = chosen;
            } else if (!have_move) {
                g_err = "no forced move for ply " + std::to_string(plies) + " before search_from";
                return 0;
            }
            gs.play(m);
            plies++;
        }
        out[0] = searched;
        written = (int64_t)w;
        return 1;
    });
    return written;
}

SP_API int64_t cattus_sp_trace_game(int game, const cattus_sp_config* c, cattus_net_eval_fn net, void* ctx, uint32_t max_plies,
                                    uint32_t* out, size_t cap) {
    return cattus_sp_trace_game_ex(game, c, net, ctx, max_plies, nullptr, 0, 0, out, cap);
}

// ---- known-answer hooks for the stochastic paths (the very functions the search calls) ----
SP_API int cattus_sp_test_dirichlet(uint64_t seed, float alpha, uint32_t k, uint32_t draws, double* out) {
    if (!out || k < 1 || !(alpha > 0)) return -1;
    Rng rng(seed);
    std::vector<double> g;
    for (uint32_t d = 0; d < draws; d++) {
        if (!sample_dirichlet(rng, alpha, k, g)) std::fill(g.begin(), g.end(), 0.0);
        std::copy(g.begin(), g.end(), out + (size_t)d * k);
    }
    return 0;
}

SP_API int cattus_sp_test_temperature_choice(uint64_t seed, const float* probs, uint32_t k, float temperature, uint32_t draws,
                                             uint32_t* counts) {
    if (!probs || !counts || k < 1 || !(temperature > 0)) return -1;
    Rng rng(seed);
    std::vector<float> w;
    std::fill(counts, counts + k, 0u);
    for (uint32_t d = 0; d < draws; d++) counts[sample_with_temperature(rng, probs, k, temperature, w)]++;
    return 0;
}

SP_API int cattus_sp_play_moves(int game, const uint16_t* moves, uint32_t n, uint32_t* plies_played) {
    int status = -100;
    dispatch(game, [&](auto g) -> int {
        typedef decltype(g) G;
        GameState<G> gs;
        gs.reset(G::Position::initial());
        std::vector<typename G::Move> legal;
        uint32_t k = 0;
        for (; k < n; k++) {
            if (gs.status().finished) break;  // play_single_turn asserts the game is ongoing (chess/core.rs:439)
            gs.history.back().legal_moves(legal);
            bool ok = false;
            for (auto& lm : legal)
                if ((uint16_t)lm.nn_idx() == moves[k]) {
                    gs.play(lm), ok = true;
                    break;
                }
            if (!ok) {
                g_err = "move " + std::to_string(moves[k]) + " is not legal at ply " + std::to_string(k);
                status = -101;
                return 0;
            }
        }
        if (plies_played) *plies_played = k;
        const Status st = gs.status();
        status = st.finished ? st.winner : 2;
        return 1;
    });
    return status;
}

SP_API cattus_pos* cattus_sp_pos_new(int game, const char* str) {
    cattus_pos* h = nullptr;
    dispatch(game, [&](auto g) -> int {
        typedef decltype(g) G;
        auto impl = std::make_unique<PosImpl<G>>();
        impl->game = game;
        if (!str) impl->p = G::Position::initial();
        else if (!parse_pos<G>(str, impl->p)) {
            g_err = "cannot parse position string";
            return 0;
        }
        h = new cattus_pos;
        h->p = std::move(impl);
        return 1;
    });
    return h;
}
SP_API void cattus_sp_pos_free(cattus_pos* p) { delete p; }
SP_API int cattus_sp_pos_status(const cattus_pos* p) { return p->p->status(); }
SP_API int cattus_sp_pos_turn(const cattus_pos* p) { return p->p->turn(); }
SP_API int cattus_sp_pos_legal(const cattus_pos* p, uint16_t* nn_idx, char* names, int cap) { return p->p->legal(nn_idx, names, cap); }
SP_API cattus_pos* cattus_sp_pos_moved(const cattus_pos* p, int k) {
    PosBase* n = p->p->moved(k);
    if (!n) return nullptr;
    auto* h = new cattus_pos;
    h->p.reset(n);
    return h;
}
SP_API cattus_pos* cattus_sp_pos_flipped(const cattus_pos* p) {
    auto* h = new cattus_pos;
    h->p.reset(p->p->flipped());
    return h;
}
SP_API int cattus_sp_pos_equal(const cattus_pos* a, const cattus_pos* b) { return a->p->equal(b->p.get()) ? 1 : 0; }
SP_API int cattus_sp_pos_planes(const cattus_pos* p, uint64_t* out) { return p->p->planes(out); }
SP_API int cattus_sp_pos_str(const cattus_pos* p, char* buf, int cap) {
    const std::string s = p->p->str();
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(buf, s.c_str(), s.size() + 1);
    return (int)s.size();
}
SP_API int cattus_sp_pos_flipped_move_nn(const cattus_pos* p, int k) { return p->p->flipped_move_nn(k); }
SP_API int cattus_sp_pos_test_record(const cattus_pos* p, uint8_t* out, int cap) { return p->p->test_record(out, cap); }
SP_API uint64_t cattus_sp_chess_perft(const char* fen, int depth) { return ChessGame::perft(ChessGame::Position::from_fen(fen), depth); }
SP_API int cattus_sp_chess_nn_moves(char* out) {
    const auto& T = chessimpl::tables();
    for (int i = 0; i < 1880; i++) {
        ChessGame::Move m{(uint8_t)T.nn_to_move[i][0], (uint8_t)T.nn_to_move[i][1], (uint8_t)T.nn_to_move[i][2]};
        const std::string s = m.str();
        memset(out + 8 * i, 0, 8);
        memcpy(out + 8 * i, s.data(), s.size());
    }
    return 1880;
}
