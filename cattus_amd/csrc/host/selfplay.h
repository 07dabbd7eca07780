// Self-play driver: many concurrent games per evaluator, leaves batched across games.
//
// Restates, on the host:
//   NNetwork::evaluate / flip / calc_moves_probs   engine/src/net/mod.rs:74-119,158-182
//   ValueFuncCache::get_or_compute                 engine/src/mcts/cache.rs:10-76
//   SelfPlayWorker::generate_data, write_data_entry  training/self-play/src/self_play.rs:179-275
//   serializers                                    training/self-play/src/self_play.rs:33-61, serialize/*.rs
//
// Threading model.  The reference runs `threads` OS threads, each playing whole games with two
// persistent MctsPlayers and meeting the other threads in Batcher::apply.  Here a "slot" is what a
// reference worker thread is (two persistent players, games pulled from a shared counter), but a
// slot is a resumable state machine: `threads` workers advance slots until their search needs a
// network evaluation, one evaluation thread gathers waiting leaves into batches and runs them
// through the `net` callback (cattus_hip_eval on the GPU), and hands the results back.
// Games, trees and evaluations are independent of how leaves are grouped into batches, so per-game
// results equal those of the reference's one-thread-per-game schedule in deterministic settings.
#pragma once
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "mcts.h"

namespace cattus {

// Raw network callback: planes [n][PLANES][PLANE_WORDS] -> policy [n][MOVES], value [n]; 0 = ok.
typedef int (*net_eval_fn)(void* ctx, const uint64_t* planes, uint32_t n, float* policy, float* value);

// Network callback that also does calc_moves_probs (net/mod.rs:100-119): legal_idx [n][stride] policy
// indices of each leaf's legal moves, legal_count [n] -> probs [n][stride], value [n]; 0 = ok.
typedef int (*net_eval_legal_fn)(void* ctx, const uint64_t* planes, uint32_t n, const uint16_t* legal_idx,
                                 const uint16_t* legal_count, uint32_t legal_stride, float* probs, float* value);

struct NetHandle {
    net_eval_fn fn = nullptr;
    void* ctx = nullptr;
    net_eval_legal_fn legal_fn = nullptr;  // when set, used instead of fn
};

struct Metrics {
    // every simulation bumps one of the first two from some search thread: keep them off each other's cache line
    alignas(64) std::atomic<uint64_t> cache_hits{0};
    alignas(64) std::atomic<uint64_t> cache_misses{0};
    alignas(64) std::atomic<uint64_t> activation_count{0};
    std::atomic<uint64_t> node_evals{0};
    std::atomic<uint64_t> plies{0};  // moves played (a game's records are counted in `positions` only when it ends)
    double run_duration_ema = 0.0, search_duration_ema = 0.0;  // RunningAverage eps 0.99 (util/metric.rs)
    std::mutex mu;
    void set_run(double s) {
        std::lock_guard<std::mutex> lk(mu);
        run_duration_ema = 0.01 * run_duration_ema + 0.99 * s;
    }
    void set_search(double s) {
        std::lock_guard<std::mutex> lk(mu);
        search_duration_ema = 0.01 * search_duration_ema + 0.99 * s;
    }
};

// ---- evaluation cache (mcts/cache.rs): key = position flipped to Player1, FIFO eviction -----------
// One FIFO of max_size positions, as the reference's single deque (cache.rs:62-66), so that evictions --
// and with them the cache.hits / cache.misses counters of the summary -- are those of a reference run with
// the same evaluation order.  The reference guards map and deque with one RwLock; every leaf of every
// search thread goes through here (hundreds of thousands per second), so the map is split into shards
// (a probe locks 1/64 of it) and the eviction order is kept by a queue whose lock is held for a push and
// a pop only -- a sleeping lock around the whole insert serialised the search threads into a convoy.
template <typename G>
class EvalCache {
   public:
    explicit EvalCache(size_t max_size) : max_size_(max_size ? max_size : 1) {}
    bool get(const typename G::Position& pos, Evaluation<G>& out) {
        const uint64_t h = pos.hash();
        Shard& s = shards_[h % SHARDS];
        std::lock_guard<std::mutex> lk(s.mu);
        auto range = s.map.equal_range(h);
        for (auto it = range.first; it != range.second; ++it)
            if (it->second.first == pos) {
                out = it->second.second;
                return true;
            }
        return false;
    }
    // returns false if the position was already present (the reference then returns the cached value)
    bool insert(const typename G::Position& pos, const Evaluation<G>& ev, Evaluation<G>* existing) {
        const uint64_t h = pos.hash();
        {
            Shard& s = shards_[h % SHARDS];
            std::lock_guard<std::mutex> lk(s.mu);
            auto range = s.map.equal_range(h);
            for (auto it = range.first; it != range.second; ++it)
                if (it->second.first == pos) {
                    if (existing) *existing = it->second.second;
                    return false;
                }
            s.map.emplace(h, std::make_pair(pos, ev));
        }
        // "remove the oldest while len >= max_size, then insert" (cache.rs:62-70): after either order of the two
        // steps the cache holds the newest max_size positions
        typename G::Position victim, entry = pos;  // copied outside the lock, moved in under it
        bool evict = false;
        {
            SpinGuard g(fifo_lock_);
            fifo_.push_back(std::move(entry));
            if (fifo_.size() > max_size_) {
                victim = std::move(fifo_.front());
                fifo_.pop_front();
                evict = true;
            }
        }
        if (evict) {
            const uint64_t oh = victim.hash();
            Shard& os = shards_[oh % SHARDS];
            std::lock_guard<std::mutex> lk(os.mu);
            auto r = os.map.equal_range(oh);
            for (auto it = r.first; it != r.second; ++it)
                if (it->second.first == victim) {
                    os.map.erase(it);
                    break;
                }
        }
        return true;
    }
    size_t size() {
        SpinGuard g(fifo_lock_);
        return fifo_.size();
    }

   private:
    struct SpinGuard {
        std::atomic_flag& f;
        // held for one push / pop: spin briefly, then give the core away (with more search threads than CPUs the holder
        // may be descheduled, and spinning through its whole quantum would be the convoy this lock replaced)
        explicit SpinGuard(std::atomic_flag& fl) : f(fl) {
            for (unsigned spins = 0; f.test_and_set(std::memory_order_acquire); spins++) {
                if (spins < 128) cpu_relax();
                else std::this_thread::yield();
            }
        }
        static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#elif defined(__aarch64__)
            asm volatile("yield");
#endif
        }
        ~SpinGuard() { f.clear(std::memory_order_release); }
    };
    static constexpr size_t SHARDS = 64;
    struct alignas(64) Shard {
        std::mutex mu;
        std::unordered_multimap<uint64_t, std::pair<typename G::Position, Evaluation<G>>> map;
    };
    Shard shards_[SHARDS];
    alignas(64) std::atomic_flag fifo_lock_ = ATOMIC_FLAG_INIT;
    std::deque<typename G::Position> fifo_;
    size_t max_size_;
};

// ---- NNetwork::evaluate split around the network call ---------------------------------------------
template <typename G>
struct PendingLeaf {
    typename G::Position pos;  // flipped so that Player1 is to move
    bool flipped = false;
    uint64_t planes[G::PLANES * G::PLANE_WORDS];
    // networks with legal_fn only: the legal moves of pos and their policy indices
    std::vector<typename G::Move> moves;
    uint16_t legal_idx[G::MOVES < 224 ? G::MOVES : 224];
    uint32_t legal_count = 0;
};

template <typename G>
class NetValueFunction {
   public:
    NetValueFunction(NetHandle net, size_t cache_size, Metrics* m) : net_(net), cache_(cache_size), metrics_(m) {}
    NetHandle net() const { return net_; }

    // flip_pos_if_needed + cache probe + to_planes (net/mod.rs:79-94). true = served from the cache.
    bool prepare(const typename G::Position& position, PendingLeaf<G>& pend, Evaluation<G>& out) {
        pend.flipped = position.turn() != PLAYER1;
        pend.pos = pend.flipped ? position.flipped() : position;
        if (cache_.get(pend.pos, out)) {
            metrics_->cache_hits++;
            unflip(out, pend.flipped);
            return true;
        }
        pend.pos.planes(pend.planes);
        if (net_.legal_fn) {
            pend.pos.legal_moves(pend.moves);
            pend.legal_count = (uint32_t)pend.moves.size();  // the driver rejects counts above the stride
            const size_t m = std::min(pend.moves.size(), sizeof pend.legal_idx / sizeof pend.legal_idx[0]);
            for (size_t i = 0; i < m; i++) pend.legal_idx[i] = (uint16_t)pend.moves[i].nn_idx();
        }
        return false;
    }
    // the same tail as finish() for a network that returned the legal-move probabilities itself
    void finish_legal(const PendingLeaf<G>& pend, const float* probs, float value, Evaluation<G>& out) {
        out.probs.clear();
        for (size_t i = 0; i < pend.moves.size(); i++) out.probs.emplace_back(pend.moves[i], probs[i]);
        out.value = value;
        if (cache_.insert(pend.pos, out, &out)) metrics_->cache_misses++;
        else metrics_->cache_hits++;
        unflip(out, pend.flipped);
    }
    // calc_moves_probs + cache insert + flip_score_if_needed (net/mod.rs:100-119, cache.rs:49-74, net/mod.rs:166-182)
    void finish(const PendingLeaf<G>& pend, const float* logits, float value, Evaluation<G>& out) {
        static thread_local std::vector<typename G::Move> moves;
        pend.pos.legal_moves(moves);
        softmax_legal(moves, logits, out.probs);
        out.value = value;
        if (cache_.insert(pend.pos, out, &out)) metrics_->cache_misses++;
        else metrics_->cache_hits++;  // someone else inserted it meanwhile: `out` now holds the cached value
        unflip(out, pend.flipped);
    }

    static void softmax_legal(const std::vector<typename G::Move>& moves, const float* logits,
                              std::vector<std::pair<typename G::Move, float>>& out) {
        out.clear();
        float max_p = -3.40282347e+38f;  // fold(f32::MIN, f32::max)
        for (auto& m : moves) {
            const float s = logits[m.nn_idx()];
            if (s > max_p) max_p = s;
        }
        float sum = 0.0f;
        for (auto& m : moves) {
            const float e = std::exp(logits[m.nn_idx()] - max_p);
            out.emplace_back(m, e);
            sum += e;
        }
        for (auto& mp : out) mp.second = mp.second / sum;
    }

   private:
    static void unflip(Evaluation<G>& ev, bool flipped) {
        if (!flipped) return;
        ev.value = -ev.value;
        for (auto& mp : ev.probs) mp.first = mp.first.flipped();
    }
    NetHandle net_;
    EvalCache<G> cache_;
    Metrics* metrics_;
};

// ---- .traindata records (self_play.rs:33-61, serialize/{chess,hex,ttt}.rs) ------------------------
template <typename G>
struct Serializer {
    static constexpr bool IS_CHESS = G::MOVES == 1880;
    static constexpr size_t RECORD_BYTES =
        IS_CHESS ? (size_t)G::PLANES * G::PLANE_WORDS * 8 + 235 + 225 * 4 + 1 : (size_t)G::PLANES * G::PLANE_WORDS * 8 + G::MOVES * 4 + 1;

    // pos must have Player1 to move; winner +1/-1/0 from Player1's side
    static void serialize(const typename G::Position& pos, std::vector<std::pair<typename G::Move, float>> probs,
                          int8_t winner, uint8_t* out) {
        uint64_t planes[G::PLANES * G::PLANE_WORDS];
        pos.planes(planes);
        uint8_t* p = out;
        memcpy(p, planes, sizeof planes);  // little-endian host
        p += sizeof planes;
        if (IS_CHESS) {
            // moves sorted by nn index; 235-byte bitmap + 225 packed probabilities (serialize/chess.rs:28-53)
            std::stable_sort(probs.begin(), probs.end(),
                             [](const auto& a, const auto& b) { return a.first.nn_idx() < b.first.nn_idx(); });
            uint8_t bitmap[235];
            float packed[225];
            memset(bitmap, 0, sizeof bitmap);
            for (auto& x : packed) x = -1.0f;
            size_t k = 0;
            for (auto& mp : probs) {
                const int idx = mp.first.nn_idx();
                bitmap[idx / 8] |= (uint8_t)(1u << (idx % 8));
                if (k < 225) packed[k++] = mp.second;
            }
            memcpy(p, bitmap, 235), p += 235;
            memcpy(p, packed, 225 * 4), p += 225 * 4;
        } else {
            float all[G::MOVES];
            for (auto& x : all) x = -1.0f;  // -1 marks illegal moves (self_play.rs:40-46)
            for (auto& mp : probs) all[mp.first.nn_idx()] = mp.second;
            memcpy(p, all, sizeof all), p += sizeof all;
        }
        *p = (uint8_t)winner;
    }
};

// Game::{new, status, play_single_turn} (game/mod.rs:70-107; chess/core.rs:405-451): the position history
// plus, for games with a repetition limit, the "a position occurred REPETITION_LIMIT times" draw flag.
template <typename G>
struct GameState {
    std::vector<typename G::Position> history;
    bool repetition_detected = false;
    void reset(const typename G::Position& start) {
        history.assign(1, start);
        repetition_detected = false;
    }
    Status status() const {  // ChessGame::status (chess/core.rs:431-436) / the position's own status
        if (repetition_detected) return Status::draw();
        return history.back().status();
    }
    void play(const typename G::Move& m) {  // play_single_turn (chess/core.rs:438-450)
        typename G::Position np = history.back().moved(m);
        if (G::REPETITION_LIMIT > 1) {
            int cnt = 1;  // seen_positions counts the new position itself
            for (auto& p : history)
                if (p == np) cnt++;
            if (cnt >= G::REPETITION_LIMIT) repetition_detected = true;
        }
        history.push_back(np);
    }
};

struct Record {
    uint32_t game_idx, pos_idx;
    uint8_t dir;  // 0 -> out_dir1, 1 -> out_dir2
    std::vector<uint8_t> bytes;
};

struct SelfPlayConfig {
    MctsParams mcts;
    uint32_t batch_size = 1;        // model.batch_size
    uint32_t threads = 1;           // host worker threads advancing slots
    size_t cache_size = 1000;       // mcts.cache_size
    uint32_t concurrent_games = 0;  // slots; 0 -> max(threads, batch_size)
    uint32_t eval_threads = 2;      // threads calling the network (batches in flight)
    // mcts.leaves_in_flight > 1 (not in the reference) lets one tree keep several leaves at the network
    uint64_t seed = 1;
    // this process plays global game indices first_game + k*game_stride, k = 0..games_num-1
    uint32_t first_game = 0, game_stride = 1;
    uint32_t max_game_plies = 0;  // > 0 (not in the reference): adjudicate a draw after this many plies
    // failure containment (include/cattus_selfplay.h): explicit game indices of a re-queue; a progress file that gets one
    // line per finished game
    std::vector<uint32_t> game_list;
    std::string progress_path;
    // optional allocator for the batch buffers handed to the network callback (page-locked memory
    // from cattus_hip_host_alloc lets the evaluator DMA straight into them)
    void* (*host_alloc)(size_t) = nullptr;
    void (*host_free)(void*) = nullptr;
};

struct SelfPlayResult {
    uint32_t w1 = 0, w2 = 0, d = 0;
    uint64_t positions = 0;
    double seconds = 0;
    // the part of the run with at least 3/4 of the slots still playing (before the drain at the end,
    // when the last games finish and batches can no longer be filled)
    double steady_seconds = 0;
    uint64_t steady_node_evals = 0;
    uint64_t steady_plies = 0, steady_batches = 0;  // moves played and batches run in that window
    uint64_t adjudicated = 0;  // games cut at max_game_plies
};

template <typename G>
class SelfPlayRunner {
   public:
    typedef typename G::Position Position;
    typedef typename G::Move Move;
    // most legal moves a position can have (chess: 218 is the known maximum); row length of the
    // legal-move buffers
    static constexpr uint32_t LEGAL_STRIDE = G::MOVES < 224 ? G::MOVES : 224;

    SelfPlayRunner(const SelfPlayConfig& cfg, NetHandle net1, NetHandle net2, bool same_model)
        : cfg_(cfg), vf1_(net1, cfg.cache_size, &metrics_), vf2_(net2, cfg.cache_size, &metrics_), same_model_(same_model) {}

    Metrics& metrics() { return metrics_; }
    const std::string& error() const { return error_; }

    // SelfPlayRunner::generate_data (self_play.rs:94-141). Records go to `records` (and to disk if
    // the out dirs are non-empty).  games_num must be even (self_play.rs:100).
    int generate_data(uint32_t games_num, const std::string& out_dir1, const std::string& out_dir2,
                      std::vector<Record>* records, SelfPlayResult& res) {
        if (games_num % 2 != 0 && cfg_.game_list.empty()) {  // a re-queued list is what was left of an even job
            error_ = "Games num should be a multiple of 2";
            return -1;
        }
        if (!cfg_.game_list.empty() && cfg_.game_list.size() != games_num) {
            error_ = "game_list must hold games_num entries";
            return -1;
        }
        progress_ = nullptr;
        if (!cfg_.progress_path.empty() && !(progress_ = fopen(cfg_.progress_path.c_str(), "a"))) {
            error_ = "cannot open progress file " + cfg_.progress_path;
            return -4;
        }
        struct ProgressCloser {
            FILE*& f;
            ~ProgressCloser() {
                if (f) fclose(f), f = nullptr;
            }
        } progress_closer{progress_};
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t nslots = cfg_.concurrent_games ? cfg_.concurrent_games : std::max(cfg_.threads, cfg_.batch_size);
        nslots = std::max(1u, std::min(nslots, std::max(games_num, 1u)));
        std::vector<Slot> slots;
        slots.reserve(nslots);
        for (uint32_t i = 0; i < nslots; i++) slots.emplace_back(cfg_.mcts, cfg_.seed * 1000003ull + i);
        std::atomic<uint32_t> next_game{0};
        std::mutex out_mu;
        const size_t words = (size_t)G::PLANES * G::PLANE_WORDS;

        // Scheduler.  `ready` holds slots the host may advance, `pending[net]` slots whose search waits
        // for a network evaluation.  Worker threads pop ready slots, consume the slot's result (if any)
        // and run its search up to the next leaf.  The evaluation thread fires a batch as soon as
        // batch_size leaves are pending, or -- when fewer are pending -- once no slot is being advanced
        // any more (then every live slot is waiting, so the batch cannot grow).  While a batch is on the
        // GPU the workers advance the other slots.  Which leaves share a batch depends on timing;
        // per-leaf results do not, so the games do not either.
        std::mutex mu;
        std::condition_variable cv_work, cv_eval;
        std::deque<uint32_t> ready;
        std::vector<uint32_t> pending[2];
        uint32_t busy = 0, done = 0;
        bool finished = false;
        int rc = 0;
        for (uint32_t i = 0; i < nslots; i++) ready.push_back(i);

        // Ring of batch buffers: a buffer is reusable once every leaf of its batch has been consumed.  A slot
        // consumes its rows only when ALL its leaves are back, and its leaves sit next to each other in the
        // pending queue, so a buffer can stay held until the ceil(leaves_in_flight / batch_size) batches behind
        // it have run: the ring must be longer than that by the batches in flight, or the evaluation threads
        // would wait for a free buffer that only their own next batch can release.
        struct BatchBuf {
            uint64_t* planes = nullptr;
            float *policy = nullptr, *value = nullptr;
            uint16_t *legal_idx = nullptr, *legal_cnt = nullptr;  // networks with legal_fn: policy holds probs
            uint32_t refs = 0;  // guarded by mu
        };
        const uint32_t lif = std::min(std::max(cfg_.mcts.leaves_in_flight, 1u), (uint32_t)MctsPlayer<G>::MAX_IN_FLIGHT);
        const uint32_t eval_threads = std::max(1u, cfg_.eval_threads);
        int NBUF = (int)std::max(6u, (lif + cfg_.batch_size - 1) / cfg_.batch_size + eval_threads + 1);
        if (!same_model_) NBUF *= 2;  // two pending queues share the ring
        std::vector<BatchBuf> bufs((size_t)NBUF);
        // Row length of the per-leaf result: all logits, or (legal_fn) one probability per legal move.
        const bool legal = vf1_.net().legal_fn != nullptr;
        if (legal != (vf2_.net().legal_fn != nullptr)) {
            error_ = "both networks must be of the same kind";
            return -4;
        }
        const size_t row = legal ? LEGAL_STRIDE : (size_t)G::MOVES;
        auto halloc = [&](size_t bytes) { return cfg_.host_alloc ? cfg_.host_alloc(bytes) : malloc(bytes); };
        auto hfree = [&](void* p) { cfg_.host_free ? cfg_.host_free(p) : free(p); };
        for (auto& b : bufs) {
            b.planes = (uint64_t*)halloc(cfg_.batch_size * words * 8);
            b.policy = (float*)halloc((size_t)cfg_.batch_size * row * 4);
            b.value = (float*)halloc(cfg_.batch_size * 4);
            if (legal) {
                b.legal_idx = (uint16_t*)halloc((size_t)cfg_.batch_size * LEGAL_STRIDE * 2);
                b.legal_cnt = (uint16_t*)halloc((size_t)cfg_.batch_size * 2);
            }
            if (!b.planes || !b.policy || !b.value || (legal && (!b.legal_idx || !b.legal_cnt))) {
                error_ = "cannot allocate batch buffers";
                return -4;
            }
        }

#ifdef CATTUS_SCHED_STATS
        // diagnostics: where the worker threads spend their time (summed over workers, seconds)
        std::atomic<uint64_t> st_wait{0}, st_lock{0}, st_adv{0}, st_iters{0};
        auto now_ns = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
#define SCHED_T(var) const uint64_t var = now_ns()
#define SCHED_ADD(acc, a, b) acc += (b) - (a)
#else
#define SCHED_T(var)
#define SCHED_ADD(acc, a, b)
#endif
        auto worker = [&]() {
            constexpr size_t CHUNK = 16;  // most slots taken per lock acquisition
            uint32_t mine[CHUNK];
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                SCHED_T(w0);
                cv_work.wait(lk, [&] { return finished || !ready.empty(); });
                SCHED_T(w1);
                SCHED_ADD(st_wait, w0, w1);
                if (finished) return;
                // an even share of what is ready, so that a burst of results spreads over all workers
                const size_t take = std::min<size_t>(CHUNK, std::max<size_t>(2, ready.size() / (2 * cfg_.threads)));
                size_t k = 0;
                while (k < take && !ready.empty()) mine[k++] = ready.front(), ready.pop_front();
                busy += (uint32_t)k;
                lk.unlock();
                // batch buffers whose rows the slots consume in this round (a slot with several leaves in flight
                // may hold rows of more than one buffer)
                int released[CHUNK][MctsPlayer<G>::MAX_IN_FLIGHT];
                uint32_t nrel[CHUNK];
                for (size_t i = 0; i < k; i++) {
                    Slot& sl = slots[mine[i]];
                    nrel[i] = 0;
                    if (sl.state == Slot::HAVE_RESULT)
                        for (uint32_t q = 0; q < sl.npend; q++)
                            if (!sl.leaf[q].cached) released[i][nrel[i]++] = sl.leaf[q].batch;
                    advance(sl, next_game, games_num, out_dir1, out_dir2, records, res, out_mu);
                }
                SCHED_T(a1);
                SCHED_ADD(st_adv, w1, a1);
                lk.lock();
                SCHED_T(l1);
                SCHED_ADD(st_lock, a1, l1);
#ifdef CATTUS_SCHED_STATS
                st_iters++;
#endif
                busy -= (uint32_t)k;
                bool wake = false;  // something an evaluation thread may be waiting for has happened
                for (size_t i = 0; i < k; i++) {
                    Slot& sl = slots[mine[i]];
                    for (uint32_t q = 0; q < nrel[i]; q++)
                        if (--bufs[released[i][q]].refs == 0) wake = true;
                    if (sl.state == Slot::WAIT_EVAL) {
                        for (uint32_t q = 0; q < sl.npend; q++)
                            if (!sl.leaf[q].cached) pending[sl.netid].push_back(mine[i] * MctsPlayer<G>::MAX_IN_FLIGHT + q);
                    } else {
                        done++, wake = true;
                    }
                }
                if (res.steady_seconds == 0 && (uint64_t)(nslots - done) * 4 < (uint64_t)nslots * 3) {
                    res.steady_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    res.steady_node_evals = metrics_.node_evals;
                    res.steady_plies = metrics_.plies, res.steady_batches = metrics_.activation_count;
                }
                if (pending[0].size() >= cfg_.batch_size || pending[1].size() >= cfg_.batch_size) wake = true;
                if (ready.empty() && busy == 0) wake = true;
                if (wake) cv_eval.notify_all();
            }
        };
        std::vector<std::thread> workers;
        for (uint32_t i = 0; i < cfg_.threads; i++) workers.emplace_back(worker);

        // Evaluation threads: each takes a batch of pending leaves through the network callback.  With two
        // of them a second batch is prepared and in flight while the first one is on the GPU (the
        // evaluator has one lane per concurrent caller).  A partial batch only goes out when nothing else
        // can add to it: no slot is being advanced and no other batch is in flight.
        uint32_t inflight = 0;
        auto evaluator = [&]() {
            std::vector<uint32_t> who;
            std::unique_lock<std::mutex> lk(mu);
            for (;;) {
                if (done == nslots || rc != 0 || finished) break;
                int netid = pending[0].size() >= pending[1].size() ? 0 : 1;
                const bool full = pending[netid].size() >= cfg_.batch_size;
                const bool drained = ready.empty() && busy == 0 && inflight == 0 && !pending[netid].empty();
                int bi = -1;
                for (int i = 0; i < NBUF; i++)
                    if (bufs[i].refs == 0) bi = i;
                if ((!full && !drained) || bi < 0) {
                    cv_eval.wait(lk);
                    continue;
                }
                BatchBuf& bb = bufs[bi];
                const size_t n = std::min<size_t>(cfg_.batch_size, pending[netid].size());
                who.assign(pending[netid].begin(), pending[netid].begin() + n);
                pending[netid].erase(pending[netid].begin(), pending[netid].begin() + n);
                bb.refs = (uint32_t)n;
                inflight++;
                lk.unlock();
                int erc = 0;
                for (size_t k = 0; k < n; k++) {
                    constexpr uint32_t MIF = MctsPlayer<G>::MAX_IN_FLIGHT;
                    const PendingLeaf<G>& pl = slots[who[k] / MIF].pend[who[k] % MIF];
                    memcpy(bb.planes + k * words, pl.planes, words * 8);
                    if (legal) {
                        if (pl.legal_count > LEGAL_STRIDE) erc = -5;
                        bb.legal_cnt[k] = (uint16_t)std::min<uint32_t>(pl.legal_count, LEGAL_STRIDE);
                        memcpy(bb.legal_idx + k * LEGAL_STRIDE, pl.legal_idx, (size_t)bb.legal_cnt[k] * 2);
                    }
                }
                NetHandle net = netid == 0 ? vf1_.net() : vf2_.net();
                const auto r0 = std::chrono::steady_clock::now();
                if (erc == 0)
                    erc = legal ? net.legal_fn(net.ctx, bb.planes, (uint32_t)n, bb.legal_idx, bb.legal_cnt, LEGAL_STRIDE, bb.policy, bb.value)
                                : net.fn(net.ctx, bb.planes, (uint32_t)n, bb.policy, bb.value);
                if (erc == 0) {
                    metrics_.set_run(std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count());
                    metrics_.activation_count++;  // counts batches, as the reference does (net/mod.rs:68)
                    metrics_.node_evals += n;
                    for (size_t k = 0; k < n; k++) {
                        constexpr uint32_t MIF = MctsPlayer<G>::MAX_IN_FLIGHT;
                        auto& lf = slots[who[k] / MIF].leaf[who[k] % MIF];
                        lf.logits = bb.policy + k * row;
                        lf.value = bb.value[k];
                        lf.batch = bi;
                    }
                }
                lk.lock();
                inflight--;
                if (erc != 0) {
                    error_ = "network evaluation failed with status " + std::to_string(erc);
                    rc = erc;
                    break;
                }
                for (size_t k = 0; k < n; k++) {  // a slot is ready again once all its leaves are back
                    Slot& sl = slots[who[k] / MctsPlayer<G>::MAX_IN_FLIGHT];
                    if (--sl.awaiting == 0) {
                        sl.state = Slot::HAVE_RESULT;
                        ready.push_back(who[k] / MctsPlayer<G>::MAX_IN_FLIGHT);
                    }
                }
                cv_work.notify_all();
            }
            finished = true;
            cv_work.notify_all();
            cv_eval.notify_all();
        };
        {
            std::vector<std::thread> evals;
            for (uint32_t i = 1; i < eval_threads; i++) evals.emplace_back(evaluator);
            evaluator();
            for (auto& t : evals) t.join();
        }
        for (auto& t : workers) t.join();
#ifdef CATTUS_SCHED_STATS
        fprintf(stderr, "sched: workers %u  idle %.2fs  lock %.2fs  advance %.2fs  iterations %llu\n", cfg_.threads, st_wait * 1e-9,
                st_lock * 1e-9, st_adv * 1e-9, (unsigned long long)st_iters.load());
#endif
        for (auto& b : bufs) {
            hfree(b.planes), hfree(b.policy), hfree(b.value);
            if (b.legal_idx) hfree(b.legal_idx), hfree(b.legal_cnt);
        }
        if (rc != 0) return rc;
        int frc = 0;
        for (auto& s : slots)
            if (!s.error.empty()) error_ = s.error, frc = -2;
        res.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (res.steady_seconds == 0)
            res.steady_seconds = res.seconds, res.steady_node_evals = metrics_.node_evals, res.steady_plies = metrics_.plies,
            res.steady_batches = metrics_.activation_count;
        return frc;
    }

   private:
    struct Slot {
        enum State { IDLE, NEXT_MOVE, SEARCHING, WAIT_EVAL, HAVE_RESULT, DONE };
        MctsPlayer<G> p1, p2;
        State state = IDLE;
        int netid = 0;
        MctsPlayer<G>* cur = nullptr;
        uint32_t game_idx = 0;
        bool players_switch = false;
        GameState<G> game;
        std::vector<std::pair<Position, std::vector<std::pair<Move, float>>>> pairs;
        // leaves of the current search waiting for an evaluation (one, unless mcts.leaves_in_flight > 1)
        PendingLeaf<G> pend[MctsPlayer<G>::MAX_IN_FLIGHT];
        struct LeafResult {
            bool cached = false;            // served from the evaluation cache: `ev` holds the result
            Evaluation<G> ev;
            const float* logits = nullptr;  // else: row (logits, or legal-move probabilities) of batch buffer `batch`
            float value = 0;
            int batch = -1;
        } leaf[MctsPlayer<G>::MAX_IN_FLIGHT];
        uint32_t npend = 0;     // leaves of the current group
        uint32_t awaiting = 0;  // of them, still at the network (guarded by the scheduler mutex)
        std::chrono::steady_clock::time_point search_t0;
        std::string error;
        Slot(const MctsParams& p, uint64_t seed) : p1(p, seed * 2 + 1), p2(p, seed * 2 + 2) {}
    };

    void advance(Slot& s, std::atomic<uint32_t>& next_game, uint32_t games_num, const std::string& d1, const std::string& d2,
                 std::vector<Record>* records, SelfPlayResult& res, std::mutex& out_mu) {
        for (;;) {
            switch (s.state) {
                case Slot::DONE:
                case Slot::WAIT_EVAL:
                    return;
                case Slot::IDLE: {
                    const uint32_t local = next_game.fetch_add(1);
                    if (local >= games_num) {
                        s.state = Slot::DONE;
                        return;
                    }
                    s.game_idx = cfg_.game_list.empty() ? cfg_.first_game + local * cfg_.game_stride : cfg_.game_list[local];
                    s.players_switch = s.game_idx % 2 == 1;
                    // the random streams belong to the game, not to the slot that happens to play it: a game
                    // is then the same whatever the schedule, the slot count or the sharding over processes
                    s.p1.reseed(mix64(cfg_.seed * 1000003ull + s.game_idx) * 2 + 1);
                    s.p2.reseed(mix64(cfg_.seed * 1000003ull + s.game_idx) * 2 + 2);
                    s.game.reset(Position::initial());
                    s.pairs.clear();
                    s.state = Slot::NEXT_MOVE;
                    break;
                }
                case Slot::NEXT_MOVE: {
                    Status st = s.game.status();
                    bool cut = false;
                    if (!st.finished && cfg_.max_game_plies && s.game.history.size() > cfg_.max_game_plies) st = Status::draw(), cut = true;
                    if (st.finished) {
                        finish_game(s, st.winner, d1, d2, records, res, out_mu, cut);
                        s.state = Slot::IDLE;
                        break;
                    }
                    Color player = s.game.history.back().turn();
                    if (s.players_switch) player = opposite(player);
                    s.cur = player == PLAYER1 ? &s.p1 : &s.p2;
                    s.netid = (player == PLAYER1 || same_model_) ? 0 : 1;
                    s.search_t0 = std::chrono::steady_clock::now();
                    s.cur->begin_search(s.game.history);
                    s.state = Slot::SEARCHING;
                    break;
                }
                case Slot::HAVE_RESULT: {
                    // every leaf of the group is resolved: deliver them in index order, whatever order they
                    // came back in, so that the tree does not depend on timing
                    auto& vf = s.netid == 0 ? vf1_ : vf2_;
                    for (uint32_t q = 0; q < s.npend; q++) {
                        if (s.leaf[q].cached) {
                            s.cur->deliver(q, s.leaf[q].ev);
                            continue;
                        }
                        Evaluation<G> ev;
                        if (vf.net().legal_fn) vf.finish_legal(s.pend[q], s.leaf[q].logits, s.leaf[q].value, ev);
                        else vf.finish(s.pend[q], s.leaf[q].logits, s.leaf[q].value, ev);
                        s.cur->deliver(q, ev);
                        s.leaf[q].logits = nullptr;  // the caller releases the batch buffer reference
                    }
                    s.npend = 0;
                    s.state = Slot::SEARCHING;
                    break;
                }
                case Slot::SEARCHING: {
                    const auto step = s.cur->advance(s.game.history);
                    if (step == MctsPlayer<G>::NEED_EVAL) {
                        auto& vf = s.netid == 0 ? vf1_ : vf2_;
                        s.npend = s.cur->pending_count();
                        uint32_t to_net = 0;
                        for (uint32_t q = 0; q < s.npend; q++) {
                            s.leaf[q].cached = vf.prepare(s.cur->pending_position(q), s.pend[q], s.leaf[q].ev);
                            if (!s.leaf[q].cached) to_net++;
                        }
                        if (to_net == 0) {
                            for (uint32_t q = 0; q < s.npend; q++) s.cur->deliver(q, s.leaf[q].ev);
                            s.npend = 0;
                        } else {
                            s.awaiting = to_net;
                            s.state = Slot::WAIT_EVAL;
                            return;
                        }
                    } else {
                        auto probs = s.cur->result();
                        metrics_.set_search(std::chrono::duration<double>(std::chrono::steady_clock::now() - s.search_t0).count());
                        Move m;
                        if (!s.cur->choose_move(s.game.history, probs, m)) {
                            s.error = "search returned no move";
                            s.state = Slot::DONE;
                            return;
                        }
                        s.pairs.emplace_back(s.game.history.back(), std::move(probs));
                        s.game.play(m);
                        metrics_.plies.fetch_add(1, std::memory_order_relaxed);
                        s.state = Slot::NEXT_MOVE;
                    }
                    break;
                }
            }
        }
    }

    // write_data_entry for every stored position + win counters (self_play.rs:219-241,248-275)
    void finish_game(Slot& s, int8_t winner, const std::string& d1, const std::string& d2, std::vector<Record>* records,
                     SelfPlayResult& res, std::mutex& out_mu, bool adjudicated = false) {
        std::vector<Record> recs;
        for (size_t pos_idx = 0; pos_idx < s.pairs.size(); pos_idx++) {
            const Position& pos = s.pairs[pos_idx].first;
            auto probs = s.pairs[pos_idx].second;
            Record r;
            r.game_idx = s.game_idx, r.pos_idx = (uint32_t)pos_idx;
            const bool p1_turn = pos.turn() == PLAYER1;
            r.dir = (uint8_t)((p1_turn ? 0 : 1) ^ (s.game_idx % 2));
            int8_t w = winner;
            Position fp = pos;
            if (!p1_turn) {  // flip_pos_if_needed + flip_score_if_needed (self_play.rs:261-263)
                fp = pos.flipped();
                w = (int8_t)-w;
                for (auto& mp : probs) mp.first = mp.first.flipped();
            }
            r.bytes.resize(Serializer<G>::RECORD_BYTES);
            Serializer<G>::serialize(fp, probs, w, r.bytes.data());
            recs.push_back(std::move(r));
        }
        std::lock_guard<std::mutex> lk(out_mu);
        bool on_disk = true;  // every record of the game reached its file whole (fclose included: that is where a full disk shows)
        for (auto& r : recs) {
            const std::string& dir = r.dir == 0 ? d1 : d2;
            if (!dir.empty()) {
                char name[64];
                snprintf(name, sizeof name, "/%08u_%03u.traindata", r.game_idx, r.pos_idx);
                FILE* f = fopen((dir + name).c_str(), "wb");
                const bool wrote = f && fwrite(r.bytes.data(), 1, r.bytes.size(), f) == r.bytes.size();
                const bool closed = f && fclose(f) == 0;
                if (!wrote || !closed) {
                    s.error = "cannot write " + dir + name;
                    on_disk = false;
                }
            }
            if (records) records->push_back(std::move(r));
        }
        res.positions += s.pairs.size();
        if (adjudicated) res.adjudicated++;
        int tally = 0;
        if (winner == 0) res.d++;
        else {
            int8_t w = winner;
            if (s.players_switch) w = (int8_t)-w;
            (w > 0 ? res.w1 : res.w2)++;
            tally = w > 0 ? 1 : 2;
        }
        // behind the game's records: a line here means the game is complete on disk -- so none is written for a game one of whose
        // files failed (the supervisor then counts it unfinished and plays it again; the run itself reports s.error)
        if (progress_ && on_disk) {
            fprintf(progress_, "%u %zu %d %d\n", s.game_idx, s.pairs.size(), tally, adjudicated ? 1 : 0);
            fflush(progress_);
        }
    }

    SelfPlayConfig cfg_;
    Metrics metrics_;
    NetValueFunction<G> vf1_, vf2_;
    bool same_model_;
    std::string error_;
    FILE* progress_ = nullptr;  // guarded by generate_data's out_mu
};

}  // namespace cattus
