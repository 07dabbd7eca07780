/*
 * cattus_selfplay.h -- C ABI of the host-side self-play library (libcattus_selfplay.so).
 *
 * It carries the reference's host half of the rollout path: game rules, PUCT search, evaluation
 * cache, the self-play driver and the .traindata serializers
 *   engine/src/mcts/{mod,cache}.rs, engine/src/net/mod.rs:74-119,158-182,
 *   engine/src/{ttt,hex,chess}/core.rs, training/self-play/src/{self_play,self_play_cmd}.rs,
 *   training/self-play/src/serialize/{chess,hex,ttt}.rs
 * and calls the network through a function pointer with the signature of cattus_hip_eval
 * (include/cattus_hip.h), so the MI355X evaluator plugs in as (fn = &cattus_hip_eval, ctx = handle).
 */
#ifndef CATTUS_SELFPLAY_H
#define CATTUS_SELFPLAY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum cattus_game {
    CATTUS_GAME_TTT = 0,
    CATTUS_GAME_HEX4 = 1,
    CATTUS_GAME_HEX5 = 2,
    CATTUS_GAME_HEX7 = 3,
    CATTUS_GAME_HEX9 = 4,
    CATTUS_GAME_HEX11 = 5,
    CATTUS_GAME_CHESS = 6,
} cattus_game;

/* planes [n][planes][plane_words] -> policy [n][moves], value [n]; returns 0 on success */
typedef int (*cattus_net_eval_fn)(void* ctx, const uint64_t* planes, uint32_t n, float* policy, float* value);

/* A network that also computes calc_moves_probs (engine/src/net/mod.rs:100-119), e.g. cattus_hip_eval_legal:
 * legal_idx [n][legal_stride] policy indices of each leaf's legal moves, legal_count [n] ->
 * probs [n][legal_stride], value [n]; returns 0 on success */
typedef int (*cattus_net_eval_legal_fn)(void* ctx, const uint64_t* planes, uint32_t n, const uint16_t* legal_idx,
                                        const uint16_t* legal_count, uint32_t legal_stride, float* probs, float* value);

/* The engine JSON of the reference (self_play_cmd.rs:34-53) plus the sharding fields. */
typedef struct cattus_sp_config {
    uint32_t struct_size;
    uint32_t sim_num;
    float explore_factor;
    uint32_t temperature_count;           /* <= 8 entries of temperature_policy */
    uint32_t temperature_threshold[8];
    float temperature_value[8];
    float prior_noise_alpha, prior_noise_epsilon;
    uint64_t cache_size;
    uint32_t batch_size;                  /* model.batch_size */
    uint32_t threads;
    uint32_t concurrent_games;            /* 0 = max(threads, batch_size) */
    uint64_t seed;
    uint32_t first_game, game_stride;     /* this process plays games first_game + k*game_stride */
    /* optional allocator for the batch buffers passed to the network callback; with
     * cattus_hip_host_alloc / cattus_hip_host_free the evaluator DMAs straight into them */
    void* (*host_alloc)(size_t);
    void (*host_free)(void*);
    /* optional: when set, player k's leaves go to legal_netk(ctxk, ...) instead of netk and the host
     * skips its own softmax; set both or neither when two networks play */
    cattus_net_eval_legal_fn legal_net1, legal_net2;
    /* threads calling the network = batches in flight (0 = 2).  libcattus_hip serves two callers
     * concurrently (one lane each), which hides transfers and launch latency behind the other batch */
    uint32_t eval_threads;
    /* 0 or 1: the reference's sequential search.  k > 1 (at most 16; not in the reference): one tree keeps up
     * to k unexpanded leaves at the network together, each holding a virtual loss on its path; fills
     * batches from few concurrent games at the price of a search that differs from the sequential one */
    uint32_t leaves_in_flight;
    /* 0: games end by the rules only (the reference).  k > 0 (not in the reference): a game still running
     * after k plies is adjudicated a draw there and its records are written as such; for bounded
     * benchmark samples of long games (summary field `adjudicated` counts them) */
    uint32_t max_game_plies;
    /* Failure containment across processes (the reference leaves a dead worker undetected: the TODO at
     * training/self-play/src/self_play.rs:128).  Both optional (NULL):
     * game_list: play exactly these global game indices (games_num entries, any parity) instead of
     *   first_game + k*game_stride -- a supervisor re-queues the unfinished games of a process that died on a fresh one;
     *   a game is a function of (seed, game index, networks), so its records come out the same wherever it is played.
     * progress_path: after ALL records of a game have been written, one line
     *   "<game_idx> <plies> <tally> <adjudicated>\n" (tally 0 draw, 1 player1_wins, 2 player2_wins -- the counter the game
     *   went to) is appended to this file and flushed: what survives of a process that dies later. */
    const uint32_t* game_list;
    const char* progress_path;
} cattus_sp_config;

typedef struct cattus_sp_summary {
    uint32_t player1_wins, player2_wins, draws;
    uint64_t positions, records;
    uint64_t activation_count;  /* batches, reference metric model.activation_count */
    uint64_t node_evals;        /* leaves sent to the network */
    uint64_t cache_hits, cache_misses;
    double run_duration, search_duration; /* reference EMA metrics model.run_duration, mcts.search_duration */
    double seconds;
    /* the run up to the moment fewer than 3/4 of the concurrent-game slots still had a game to play,
     * i.e. without the drain at the end when batches can no longer be filled */
    double steady_seconds;
    uint64_t steady_node_evals;
    uint64_t adjudicated;       /* games cut at max_game_plies (counted among the draws) */
    uint64_t steady_plies;      /* moves played in the steady window (positions counts a game's records only when it ends) */
    uint64_t steady_batches;    /* batches run in the steady window: steady_node_evals / steady_batches = its batch fill */
} cattus_sp_summary;

typedef struct cattus_sp_result cattus_sp_result;

/* game geometry: out = {board, moves, planes, plane_words, record_bytes} */
int cattus_sp_game_info(int game, uint32_t out[5]);

/* run_main (self_play_cmd.rs:55-153) minus argument parsing: plays games_num games, writes one
 * .traindata file per position into out_dir1/out_dir2 when they are non-NULL, keeps the records in the
 * result when keep_records != 0.  net2 == NULL means both players share net1 (and its cache). */
int cattus_sp_run(int game, const cattus_sp_config* cfg, cattus_net_eval_fn net1, void* ctx1, cattus_net_eval_fn net2,
                  void* ctx2, uint32_t games_num, const char* out_dir1, const char* out_dir2, int keep_records,
                  cattus_sp_result** out);
int cattus_sp_result_summary(const cattus_sp_result* r, cattus_sp_summary* out);
/* bytes: [records][record_bytes]; meta: [records][3] = game_idx, pos_idx, dir(0/1) */
int cattus_sp_result_records(const cattus_sp_result* r, uint8_t* bytes, uint32_t* meta);
void cattus_sp_result_free(cattus_sp_result* r);
const char* cattus_sp_last_error(void);

/* Deterministic stand-in network (tests, plumbing runs): logits and value are hashes of the planes.
 * ctx must point to uint32_t[2] = {moves, plane words per leaf}. */
int cattus_sp_stub_net(void* ctx, const uint64_t* planes, uint32_t n, float* policy, float* value);

/* Known-answer hook: one game of self-play with the given network, returning for every ply the root
 * visit counts in result order.  out: [ply count, then per ply: chosen nn_idx, k, k x (nn_idx, visits)].
 * Returns the number of uint32 written, or a negative status. */
int64_t cattus_sp_trace_game(int game, const cattus_sp_config* cfg, cattus_net_eval_fn net, void* ctx, uint32_t max_plies,
                             uint32_t* out, size_t cap);

/* The same with a forced line: plies < n_forced play forced[ply] (policy indices; must be legal) whatever the
 * search chose, and plies < search_from are played without a search at all (an opening).  Every searched ply
 * is still reported with the move the search chose.  max_plies bounds the searched plies.  Used to search the
 * same positions with two evaluators, and to put a repetition in front of a search. */
int64_t cattus_sp_trace_game_ex(int game, const cattus_sp_config* cfg, cattus_net_eval_fn net, void* ctx, uint32_t max_plies,
                                const uint16_t* forced, uint32_t n_forced, uint32_t search_from, uint32_t* out, size_t cap);

/* Game::play_single_turn over a list of moves (policy indices) from the initial position, with the game-level
 * rules on top of the position's: chess threefold repetition (chess/core.rs:438-450).  Stops at the first
 * finished state.  Returns 2 = ongoing, +1 / -1 / 0 = winner / draw, <= -100 = error; *plies_played = moves
 * applied. */
int cattus_sp_play_moves(int game, const uint16_t* moves, uint32_t n, uint32_t* plies_played);

/* Known-answer hooks for the stochastic paths, running the very functions the search calls:
 * `draws` samples of the root-noise distribution Dir(alpha, ..., alpha) over k moves (mcts/mod.rs:419-446,
 * util/dirichlet.rs:226-352) into out[draws][k], and `draws` move choices at a temperature > 0 among k moves with
 * visit probabilities probs[k] (mcts/mod.rs:403-415: chosen with probability ~ p^(1/T)) tallied into counts[k]. */
int cattus_sp_test_dirichlet(uint64_t seed, float alpha, uint32_t k, uint32_t draws, double* out);
int cattus_sp_test_temperature_choice(uint64_t seed, const float* probs, uint32_t k, float temperature, uint32_t draws,
                                      uint32_t* counts);

/* ---- position handles (rule tests) ---------------------------------------------------------- */
typedef struct cattus_pos cattus_pos;
/* str: NULL = initial position; ttt "xo_..."+turn, hex "reb..."+turn (test_util.rs:7-66), chess FEN */
cattus_pos* cattus_sp_pos_new(int game, const char* str);
void cattus_sp_pos_free(cattus_pos* p);
int cattus_sp_pos_status(const cattus_pos* p);            /* 2 ongoing; else winner +1 / -1 / 0 */
int cattus_sp_pos_turn(const cattus_pos* p);              /* 0 Player1, 1 Player2 */
int cattus_sp_pos_legal(const cattus_pos* p, uint16_t* nn_idx, char* names /* [cap][12] */, int cap);
cattus_pos* cattus_sp_pos_moved(const cattus_pos* p, int kth_legal_move);
cattus_pos* cattus_sp_pos_flipped(const cattus_pos* p);
int cattus_sp_pos_equal(const cattus_pos* a, const cattus_pos* b);
int cattus_sp_pos_planes(const cattus_pos* p, uint64_t* out);
int cattus_sp_pos_str(const cattus_pos* p, char* buf, int cap);
int cattus_sp_pos_flipped_move_nn(const cattus_pos* p, int kth_legal_move); /* nn_idx of move.flipped() */
/* test_serialize.rs:63-79: probs idx/(n*(n-1)), winner by n%3, serialised record of the position */
int cattus_sp_pos_test_record(const cattus_pos* p, uint8_t* out, int cap);
uint64_t cattus_sp_chess_perft(const char* fen, int depth);
/* the 1880 policy-index moves as LAN strings, 8 bytes each (chess/core.rs:453-593) */
int cattus_sp_chess_nn_moves(char* out /* [1880][8] */);

#ifdef __cplusplus
}
#endif
#endif
