// PUCT Monte-Carlo tree search, restated from the reference engine/src/mcts/mod.rs:58-489 so that,
// given the same leaf evaluations, it visits the same nodes in the same order.
//
// The reference search blocks inside ValueFunction::evaluate (mcts/mod.rs:264-268).  Here the
// search is resumable instead: advance() runs simulations until one needs a network evaluation and
// returns NEED_EVAL; the caller evaluates the pending leaf (typically batched with the leaves of
// hundreds of other games on the GPU) and calls deliver().  Simulations of one tree stay strictly
// sequential, exactly as in the reference (no virtual loss), so visit counts are unchanged.
//
// Iteration-order semantics carried over from the reference's graph library (petgraph 0.8.3, not in
// the reference tree):
//   * edges(node) yields the most recently added edge first -> children are stored in insertion order
//     and walked in reverse;
//   * Iterator::max_by keeps the LAST maximal element -> ties go to the earliest-inserted child;
//   * remove_all_but_subtree re-adds children in edges() order -> sibling order reverses on every
//     tree reuse (mcts/mod.rs:303-333).
//
// Child nodes are materialised lazily.  The reference's create_children makes one node (a moved position)
// per legal move at once (mcts/mod.rs:246-262); at 800 simulations per move only ~1 in 30 of them is ever
// visited.  Here an expansion creates the EDGES only (move, prior, n, w -- everything selection and the
// result read); the child's node, i.e. its position, is made when a selection first walks the edge.
// Nothing observable depends on when a node is made: selection, backup and the visit counts read edges,
// repetition detection reads the positions of the selected path (materialised by the selection), and the
// tree-reuse lookup compares positions in the reference's breadth-first order, computing those of
// unmaterialised children on the fly.  Trees are ~20x smaller and an expansion costs one make-move, not 30.
#pragma once
#include <cassert>
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

#include "games.h"

namespace cattus {

// splitmix64 stream: the only randomness of the host side (temperature sampling, Dirichlet noise).
// The reference uses an unseeded thread RNG there (mcts/mod.rs:415,435), so those paths are not
// reproducible in the reference either; parity runs use temperature 0 and noise off.
struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed = 0x1234) : s(seed) {}
    uint64_t next() {
        s += 0x9E3779B97F4A7C15ull;
        return mix64(s);
    }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }  // [0,1)
    double normal() {
        double u1 = uniform(), u2 = uniform();
        if (u1 < 1e-300) u1 = 1e-300;
        return std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    }
    double gamma(double alpha) {  // Marsaglia-Tsang; alpha < 1 via the boost trick
        if (alpha < 1.0) {
            double u = uniform();
            if (u < 1e-300) u = 1e-300;
            return gamma(alpha + 1.0) * std::pow(u, 1.0 / alpha);
        }
        const double d = alpha - 1.0 / 3.0, c = 1.0 / std::sqrt(9.0 * d);
        for (;;) {
            double x = normal(), v = 1.0 + c * x;
            if (v <= 0) continue;
            v = v * v * v;
            const double u = uniform();
            if (u < 1.0 - 0.0331 * x * x * x * x) return d * v;
            if (std::log(u) < 0.5 * x * x + d * (1.0 - v + std::log(v))) return d * v;
        }
    }
};

// One draw from the symmetric Dirichlet distribution Dir(alpha, ..., alpha) over k categories: k Gamma(alpha, 1) variates
// over their sum (what util/dirichlet.rs:226-352 samples for mcts/mod.rs:435-441).  Returns false when the sum is not
// positive (every variate underflowed: the reference redraws on non-finite noise; here the priors stay as they are).
inline bool sample_dirichlet(Rng& rng, double alpha, size_t k, std::vector<double>& out) {
    out.resize(k);
    double sum = 0;
    for (auto& x : out) {
        x = rng.gamma(alpha);
        sum += x;
    }
    if (!(sum > 0)) return false;
    for (auto& x : out) x /= sum;
    return true;
}

// Index drawn with probability proportional to p_i^(1/t) (mcts/mod.rs:403-415: powf, f32 sum, WeightedIndex).
inline size_t sample_with_temperature(Rng& rng, const float* probs, size_t n, float t, std::vector<float>& w) {
    w.resize(n);
    float sum = 0.0f;
    for (size_t i = 0; i < n; i++) {
        w[i] = std::pow(probs[i], 1.0f / t);
        sum += w[i];
    }
    const double u = rng.uniform() * (double)sum;
    double accw = 0.0;
    for (size_t i = 0; i < n; i++) {
        accw += w[i];
        if (u < accw) return i;
    }
    return n - 1;
}

// TemperaturePolicy (mcts/mod.rs:456-489)
struct TemperaturePolicy {
    std::vector<std::pair<size_t, float>> scheduled;  // strictly increasing thresholds
    float last = 1.0f;
    float get(size_t move_num) const {
        for (auto& t : scheduled)
            if (move_num < t.first) return t.second;
        return last;
    }
};

struct MctsParams {  // mcts/mod.rs:72-79
    uint32_t sim_num = 100;
    float explore_factor = 1.41421356f;
    TemperaturePolicy temperature;
    float prior_noise_alpha = 0.0f, prior_noise_epsilon = 0.0f;
    // Not in the reference (its search is strictly sequential): up to this many unexpanded leaves of ONE
    // tree may wait for the network together, each holding a virtual loss on its path so that the next
    // selection goes elsewhere.  1 = the reference's search, simulation by simulation.
    uint32_t leaves_in_flight = 1;
};

template <typename G>
struct Evaluation {  // what ValueFunction::evaluate returns (mcts/value_func.rs:1-11)
    std::vector<std::pair<typename G::Move, float>> probs;
    float value = 0.0f;
};

template <typename G>
class MctsPlayer {
   public:
    typedef typename G::Position Position;
    typedef typename G::Move Move;
    enum Step { NEED_EVAL, SEARCH_DONE };

    explicit MctsPlayer(const MctsParams& p, uint64_t seed = 1) : params_(p), rng_(seed) {
        assert(p.sim_num > 0 && p.explore_factor >= 0 && p.prior_noise_alpha >= 0);
    }

    // restart the random stream (temperature sampling, Dirichlet noise)
    void reseed(uint64_t seed) { rng_ = Rng(seed); }

    // calc_moves_probabilities, first half (mcts/mod.rs:335-358): tree reuse, root creation.
    void begin_search(const std::vector<Position>& history) {
        assert(params_.sim_num > 1);  // develop_tree asserts this (mcts/mod.rs:157)
        const Position& position = history.back();
        if (has_root_) {
            const int node = find_node_with_position(position, 3);
            if (node >= 0) {
                remove_all_but_subtree((uint32_t)node);
            } else {  // not in the tree, or a child nobody visited: either way the new tree is the bare root
                nodes_.clear();
                edges_.clear();
                has_root_ = false;
            }
        }
        if (!has_root_) {
            nodes_.push_back(Node(position, 0, 0));
            root_ = 0;
            has_root_ = true;
        }
        assert(position == nodes_[root_].pos);
        index_history(history);
        sims_done_ = 0;
        inflight_n_ = delivered_ = 0;
    }

    // develop_tree (mcts/mod.rs:156-196), suspended at the evaluation of unexpanded leaves.  Returns
    // NEED_EVAL with pending_count() leaves to evaluate (always 1 unless leaves_in_flight > 1); the caller
    // delivers every one of them before calling advance() again.
    static constexpr uint32_t MAX_IN_FLIGHT = 16;
    Step advance(const std::vector<Position>& history) {
        (void)history;
        assert(inflight_n_ == 0);
        const uint32_t K = std::min(std::max(params_.leaves_in_flight, 1u), MAX_IN_FLIGHT);
        while (sims_done_ + inflight_n_ < params_.sim_num && inflight_n_ < K) {
            InFlight& f = inflight_[inflight_n_];
            select(f.path);
            const uint32_t leaf = f.path.empty() ? root_ : edges_[f.path.back()].target;  // made by select()
            if (nodes_[leaf].pending) break;  // selection ran into a leaf already waiting: evaluate what we have
            const bool repetition = detect_repetition(f.path);
            const Status st = nodes_[leaf].status();
            if (repetition) {
                backpropagate(f.path, 0.0f);
                sims_done_++;
            } else if (st.finished) {
                backpropagate(f.path, (float)st.winner);
                sims_done_++;
            } else {
                f.leaf = leaf;
                nodes_[leaf].pending = 1;
                if (K > 1)
                    for (uint32_t e : f.path) edges_[e].vl++;
                inflight_n_++;
            }
        }
        delivered_ = 0;
        return inflight_n_ ? NEED_EVAL : SEARCH_DONE;
    }
    uint32_t pending_count() const { return inflight_n_; }
    const Position& pending_position(uint32_t i = 0) const { return nodes_[inflight_[i].leaf].pos; }

    // second half of a simulation: create_children, root noise, backpropagate (mcts/mod.rs:179-194)
    // Returns false (and changes nothing) for an index that is not waiting: out of range or delivered before.
    bool deliver(const Evaluation<G>& ev) { return deliver(0, ev); }
    bool deliver(uint32_t i, const Evaluation<G>& ev) {
        if (i >= inflight_n_) return false;
        const InFlight& f = inflight_[i];
        if (nodes_[f.leaf].pending == 0) return false;  // delivered before: every leaf counts once
        nodes_[f.leaf].pending = 0;
        for (uint32_t e : f.path)
            if (edges_[e].vl) edges_[e].vl--;
        create_children(f.leaf, ev.probs);
        if (f.leaf == root_) add_dirichlet_noise(root_);
        backpropagate(f.path, ev.value);
        sims_done_++;
        if (++delivered_ == inflight_n_) inflight_n_ = 0;
        return true;
    }

    // calc_moves_probabilities, second half (mcts/mod.rs:363-379): (move, n / sum n) in edges() order
    std::vector<std::pair<Move, float>> result() const {
        std::vector<std::pair<Move, float>> res;
        const Node& r = nodes_[root_];
        uint32_t total = 0;
        for (uint32_t i = r.count; i-- > 0;) total += edges_[r.first + i].n;
        for (uint32_t i = r.count; i-- > 0;) {
            const Edge& e = edges_[r.first + i];
            res.emplace_back(e.m, (float)e.n / (float)total);
        }
        return res;
    }
    // raw visit counts in the same order (for tests)
    std::vector<std::pair<Move, uint32_t>> root_visits() const {
        std::vector<std::pair<Move, uint32_t>> res;
        const Node& r = nodes_[root_];
        for (uint32_t i = r.count; i-- > 0;) res.emplace_back(edges_[r.first + i].m, edges_[r.first + i].n);
        return res;
    }

    // choose_move_from_probabilities (mcts/mod.rs:387-417)
    bool choose_move(const std::vector<Position>& history, const std::vector<std::pair<Move, float>>& probs, Move& out) {
        if (probs.empty()) return false;
        const float t = params_.temperature.get(history.size() / 2);
        if (t == 0.0f) {
            size_t best = 0;  // max_by(total_cmp): the last maximal element wins
            for (size_t i = 1; i < probs.size(); i++)
                if (!(probs[best].second > probs[i].second)) best = i;
            out = probs[best].first;
            return true;
        }
        std::vector<float> p(probs.size()), w;
        for (size_t i = 0; i < probs.size(); i++) p[i] = probs[i].second;
        out = probs[sample_with_temperature(rng_, p.data(), p.size(), t, w)].first;
        return true;
    }

    void clear() {
        nodes_.clear();
        edges_.clear();
        has_root_ = false;
        inflight_n_ = delivered_ = 0;
    }
    size_t tree_nodes() const { return nodes_.size(); }

   private:
    struct Node {
        Position pos;
        uint32_t first, count;  // children edges [first, first+count) in insertion order
        // Position::status() memoised on first use: the reference re-derives it on every visit
        // (mcts/mod.rs:207), which for chess means a move generation per visited node per simulation.
        mutable int8_t st_known = 0, st_finished = 0, st_winner = 0;
        uint8_t pending = 0;  // an evaluation of this (unexpanded) node is in flight
        Node(const Position& p, uint32_t f, uint32_t c) : pos(p), first(f), count(c) {}
        Status status() const {
            if (!st_known) {
                const Status s = pos.status();
                st_known = 1, st_finished = s.finished, st_winner = s.winner;
            }
            return Status{(bool)st_finished, st_winner};
        }
    };
    static constexpr uint32_t NO_NODE = 0xffffffffu;
    struct Edge {  // MctsEdge (mcts/mod.rs:32-45) + endpoints (target NO_NODE until a selection walks the edge)
        Move m;
        float init_score;
        uint32_t n;
        float w;
        uint32_t source, target;
        uint32_t vl = 0;  // virtual losses: in-flight simulations through this edge (0 in the reference's search)
    };

    // calc_selection_heuristic (mcts/mod.rs:233-244), f32 throughout
    float heuristic(const Edge& e, uint32_t parent_simcount) const {
        // an in-flight simulation counts as a visit that lost; with vl == 0 this is the reference's formula
        const uint32_t n = e.n + e.vl;
        const float w = e.vl ? e.w - (float)e.vl : e.w;
        const float exploit = n == 0 ? 0.0f : w / (float)n;
        const float explore = params_.explore_factor * e.init_score * (std::sqrt((float)parent_simcount) / (float)(1 + n));
        return exploit + explore;
    }

    void select(std::vector<uint32_t>& path) {  // mcts/mod.rs:199-231
        path.clear();
        uint32_t node_id = root_;
        for (;;) {
            const Node& node = nodes_[node_id];
            if (node.count == 0 || node.status().finished) return;
            uint32_t simcount = 1;
            for (uint32_t i = 0; i < node.count; i++) simcount += edges_[node.first + i].n + edges_[node.first + i].vl;
            // edges() order = newest first; max_by keeps y unless cmp(best, y) == Greater
            uint32_t best = node.first + node.count - 1;
            float vbest = heuristic(edges_[best], simcount);
            for (uint32_t i = node.count - 1; i-- > 0;) {
                const uint32_t e = node.first + i;
                const float v = heuristic(edges_[e], simcount);
                if (!(vbest > v)) {
                    best = e;
                    vbest = v;
                }
            }
            path.push_back(best);
            node_id = target_node(best);
        }
    }

    // the node an edge leads to, made on first use (see the note on lazy nodes at the top of the file)
    uint32_t target_node(uint32_t e) {
        if (edges_[e].target == NO_NODE) {
            const uint32_t child = (uint32_t)nodes_.size();
            Position np = nodes_[edges_[e].source].pos.moved(edges_[e].m);
            nodes_.push_back(Node(np, 0, 0));
            edges_[e].target = child;
        }
        return edges_[e].target;
    }

    // mcts/mod.rs:133-154: count equal positions over game history + search path; the first position
    // whose count reaches the limit makes the leaf a draw.  The game history is fixed during one
    // search, so its positions are counted once (begin_search) instead of once per simulation.
    void index_history(const std::vector<Position>& history) {
        hist_.clear();
        hist_limit_reached_ = false;
        if (G::REPETITION_LIMIT <= 1) return;
        for (const Position& p : history) {
            const uint64_t h = p.hash();
            bool found = false;
            for (auto& e : hist_)
                if (e.hash == h && *e.pos == p) {
                    if (++e.count >= G::REPETITION_LIMIT) hist_limit_reached_ = true;
                    found = true;
                    break;
                }
            if (!found) {
                hist_.push_back(HistEntry{h, &p, 1});
                if (1 >= G::REPETITION_LIMIT) hist_limit_reached_ = true;
            }
        }
    }
    bool detect_repetition(const std::vector<uint32_t>& path) const {
        if (G::REPETITION_LIMIT <= 1) return false;
        if (hist_limit_reached_) return true;
        for (size_t i = 0; i < path.size(); i++) {
            const Position& p = nodes_[edges_[path[i]].target].pos;
            const uint64_t h = p.hash();
            int cnt = 1;
            for (const auto& e : hist_)
                if (e.hash == h && *e.pos == p) {
                    cnt += e.count;
                    break;
                }
            for (size_t j = 0; j < i; j++) {
                const Position& q = nodes_[edges_[path[j]].target].pos;
                if (q.hash() == h && q == p) cnt++;
            }
            if (cnt >= G::REPETITION_LIMIT) return true;
        }
        return false;
    }

    void create_children(uint32_t parent, const std::vector<std::pair<Move, float>>& per_move) {  // mcts/mod.rs:246-262
        nodes_[parent].first = (uint32_t)edges_.size();
        nodes_[parent].count = (uint32_t)per_move.size();
        for (auto& mp : per_move) edges_.push_back(Edge{mp.first, mp.second, 0, 0.0f, parent, NO_NODE});
    }

    void backpropagate(const std::vector<uint32_t>& path, float score) {  // mcts/mod.rs:270-281
        for (uint32_t e : path) {
            Edge& edge = edges_[e];
            edge.n += 1;
            edge.w += nodes_[edge.source].pos.turn() == PLAYER1 ? score : -score;
        }
    }

    // mcts/mod.rs:283-301: breadth-first over `depth_limit` layers, children in edges() order, first equal
    // position wins.  Returns the node, FOUND_UNMADE if the first match is a child whose node was never
    // made (it has no subtree: the caller starts from a single root), or -1.
    static constexpr int FOUND_UNMADE = -2;
    int find_node_with_position(const Position& position, uint32_t depth_limit) const {
        struct Ref {
            uint32_t node;  // NO_NODE: the unmade target of `edge`
            uint32_t edge;
        };
        std::vector<Ref> layer{Ref{root_, 0}}, next;
        for (uint32_t d = 0; d < depth_limit; d++) {
            next.clear();
            for (const Ref& r : layer) {
                if (r.node == NO_NODE) {
                    const Edge& e = edges_[r.edge];
                    if (nodes_[e.source].pos.moved(e.m) == position) return FOUND_UNMADE;
                    continue;  // no node, no children
                }
                if (nodes_[r.node].pos == position) return (int)r.node;
                const Node& nd = nodes_[r.node];
                for (uint32_t i = nd.count; i-- > 0;) next.push_back(Ref{edges_[nd.first + i].target, nd.first + i});
            }
            layer.swap(next);
        }
        return -1;
    }

    void remove_all_but_subtree(uint32_t sub_root) {  // mcts/mod.rs:303-333
        if (root_ == sub_root) return;
        // the copy goes into a second pair of vectors kept by the player: no allocation once they have grown
        std::vector<Node>& nn = nodes2_;
        std::vector<Edge>& ne = edges2_;
        nn.clear();
        ne.clear();
        nn.push_back(nodes_[sub_root]);
        std::vector<std::pair<uint32_t, uint32_t>>& stack = copy_stack_;
        stack.clear();
        stack.emplace_back(sub_root, 0u);
        while (!stack.empty()) {
            const auto [po, pn] = stack.back();
            stack.pop_back();
            const Node& old = nodes_[po];
            nn[pn].first = (uint32_t)ne.size();
            nn[pn].count = old.count;
            for (uint32_t i = old.count; i-- > 0;) {  // edges() order; re-added in that order
                const Edge& e = edges_[old.first + i];
                if (e.target == NO_NODE) {
                    ne.push_back(Edge{e.m, e.init_score, e.n, e.w, pn, NO_NODE});
                    continue;
                }
                const uint32_t cn = (uint32_t)nn.size();
                nn.push_back(nodes_[e.target]);
                ne.push_back(Edge{e.m, e.init_score, e.n, e.w, pn, cn});
                stack.emplace_back(e.target, cn);
            }
        }
        nodes_.swap(nn);
        edges_.swap(ne);
        root_ = 0;
        if (nodes_[root_].count > 0) add_dirichlet_noise(root_);
    }

    void add_dirichlet_noise(uint32_t node_id) {  // mcts/mod.rs:419-446
        if (params_.prior_noise_alpha == 0.0f || params_.prior_noise_epsilon == 0.0f) return;
        const Node& nd = nodes_[node_id];
        if (nd.count < 2) return;
        std::vector<double> g;
        if (!sample_dirichlet(rng_, params_.prior_noise_alpha, nd.count, g)) return;
        const float eps = params_.prior_noise_epsilon;
        uint32_t k = 0;
        for (uint32_t i = nd.count; i-- > 0; k++) {  // edges() order
            Edge& e = edges_[nd.first + i];
            e.init_score = (1.0f - eps) * e.init_score + eps * (float)g[k];
        }
    }

    MctsParams params_;
    Rng rng_;
    std::vector<Node> nodes_, nodes2_;
    std::vector<Edge> edges_, edges2_;
    std::vector<std::pair<uint32_t, uint32_t>> copy_stack_;
    uint32_t root_ = 0;
    bool has_root_ = false;

    // suspended-simulation state
    uint32_t sims_done_ = 0;
    struct InFlight {
        std::vector<uint32_t> path;
        uint32_t leaf = 0;
    };
    InFlight inflight_[MAX_IN_FLIGHT];
    uint32_t inflight_n_ = 0, delivered_ = 0;
    struct HistEntry {
        uint64_t hash;
        const Position* pos;  // into the caller's history vector, which outlives the search
        int count;
    };
    std::vector<HistEntry> hist_;
    bool hist_limit_reached_ = false;
};

}  // namespace cattus
