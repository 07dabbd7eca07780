"""Pure-Python restatement of the reference's search + evaluator glue (TEST INFRASTRUCTURE ONLY).

An independent second implementation, written from the reference text, against which the C++ host
library (cattus_amd/csrc/host) is cross-checked on small games:

  MctsPlayer                      engine/src/mcts/mod.rs:58-446
  NNetwork::evaluate / flip / softmax   engine/src/net/mod.rs:74-119,158-182
  ValueFuncCache                  engine/src/mcts/cache.rs:31-75 (as a dict; FIFO irrelevant at test sizes)
  tic-tac-toe, Hex rules          engine/src/ttt/core.rs, engine/src/hex/core.rs
  repetition (in search, in game) engine/src/mcts/mod.rs:133-154, engine/src/chess/core.rs:438-450
  self-play game loop             training/self-play/src/self_play.rs:179-217

The search tree mimics petgraph's adjacency lists literally: every node keeps its outgoing edges in
insertion order and ``edges()`` walks them newest-first; ``max_by`` keeps the last maximum.
All search arithmetic is done in numpy float32, as the reference does in f32.
"""

from __future__ import annotations

import numpy as np

from oracle import oracle

F = np.float32
MASK = (1 << 64) - 1


def mix64(z: int) -> int:
    z &= MASK
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK
    return z ^ (z >> 31)


def stub_net(planes_words, moves: int):
    """Same deterministic stand-in network as cattus_sp_stub_net (include/cattus_selfplay.h)."""
    h = 0x243F6A8885A308D3
    for w in planes_words:
        h = mix64(h ^ int(w))
    logits = np.empty(moves, dtype=np.float32)
    for m in range(moves):
        x = mix64(h ^ (((m + 1) * 0x9E3779B97F4A7C15) & MASK))
        logits[m] = F(x >> 40) * F(1.0 / 16777216.0) * F(4.0) - F(2.0)
    x = mix64(h ^ 0xABCDEF0123456789)
    value = F(x >> 40) * F(1.0 / 16777216.0) * F(2.0) - F(1.0)
    return logits, value


# ------------------------------------------------------------------------------------- games


class Ttt:
    MOVES, REPETITION_LIMIT = 9, None
    LINES = [0b111000000, 0b000111000, 0b000000111, 0b100100100, 0b010010010, 0b001001001, 0b100010001, 0b001010100]

    def __init__(self, x=0, o=0, turn=1):
        self.x, self.o, self.turn = x, o, turn  # turn: 1 = Player1, 2 = Player2
        self.winner = None
        for line in self.LINES:  # x checked before o per line (ttt/core.rs:178-189)
            if self.x & line == line:
                self.winner = 1
                break
            if self.o & line == line:
                self.winner = 2
                break

    def key(self):
        return (self.x, self.o, self.turn)

    def status(self):
        if self.winner:
            return ("finished", self.winner)
        if (self.x | self.o) == 0x1FF:
            return ("finished", None)
        return ("ongoing", None)

    def legal_moves(self):
        return [i for i in range(9) if not ((self.x | self.o) >> i) & 1]

    def moved(self, m):
        if self.turn == 1:
            return Ttt(self.x | (1 << m), self.o, 2)
        return Ttt(self.x, self.o | (1 << m), 1)

    def flipped(self):
        return Ttt(self.o, self.x, 3 - self.turn)

    @staticmethod
    def flip_move(m):
        return m

    def planes(self):
        return [self.x, self.o, 0x1FF]


def make_hex(n: int):
    class Hex:
        N, MOVES, REPETITION_LIMIT = n, n * n, None
        DIRS = [(0, 1), (-1, 0), (-1, -1), (0, -1), (1, 0), (1, 1)]

        def __init__(self, red=0, blue=0, turn=1):
            self.red, self.blue, self.turn = red, blue, turn

        def key(self):
            return (self.red, self.blue, self.turn)

        def _connected(self, board, player):
            # red (Player1) joins column 0 to column N-1, blue row 0 to row N-1 (hex/core.rs:117-124,218-270)
            seen, stack = set(), []
            for i in range(n):
                idx = i * n if player == 1 else i
                if board >> idx & 1:
                    stack.append(idx)
                    seen.add(idx)
            while stack:
                idx = stack.pop()
                r, c = divmod(idx, n)
                if (c if player == 1 else r) == n - 1:
                    return True
                for dr, dc in self.DIRS:
                    nr, nc = r + dr, c + dc
                    if 0 <= nr < n and 0 <= nc < n:
                        j = nr * n + nc
                        if j not in seen and board >> j & 1:
                            seen.add(j)
                            stack.append(j)
            return False

        def status(self):
            if self._connected(self.red, 1):
                return ("finished", 1)
            if self._connected(self.blue, 2):
                return ("finished", 2)
            if bin(self.red | self.blue).count("1") == n * n:
                return ("finished", None)
            return ("ongoing", None)

        def legal_moves(self):
            occ = self.red | self.blue
            return [i for i in range(n * n) if not occ >> i & 1]

        def moved(self, m):
            if self.turn == 1:
                return Hex(self.red | (1 << m), self.blue, 2)
            return Hex(self.red, self.blue | (1 << m), 1)

        @staticmethod
        def _t(b):
            out = 0
            for r in range(n):
                for c in range(n):
                    if b >> (r * n + c) & 1:
                        out |= 1 << (c * n + r)
            return out

        def flipped(self):
            return Hex(self._t(self.blue), self._t(self.red), 3 - self.turn)

        @staticmethod
        def flip_move(m):
            return (m % n) * n + m // n

        def planes(self):
            full = (1 << (n * n)) - 1
            out = []
            for p in (self.red, self.blue, full):
                out += [p & MASK, p >> 64]
            return out

    return Hex


_CHESS_NAMES = None


def make_chess():
    """Chess for the search restatement.  The RULES (legal moves in the reference's order, make-move, status,
    flip, planes) come from the host library's position handles -- they have their own tests
    (tests/test_host_rules.py: perft, the reference's mate / fifty-move / flip cases); what is restated
    independently here is everything the search and the game loop do WITH positions: equality that ignores
    the fifty-move counter (chess/core.rs:288-305; the FEN without counters is the key), the in-search
    repetition rule (mcts/mod.rs:133-154) and the game-level threefold rule (chess/core.rs:438-450)."""
    from cattus_amd import selfplay as sp

    global _CHESS_NAMES
    if _CHESS_NAMES is None:
        names = sp.chess_nn_moves()
        _CHESS_NAMES = (names, {nm: i for i, nm in enumerate(names)})
    names, index = _CHESS_NAMES

    class Chess:
        MOVES, REPETITION_LIMIT = 1880, 3

        def __init__(self, h=None):
            self.h = h if h is not None else sp.Position("chess")
            self.turn = 1 if self.h.turn() == 0 else 2
            self._legal = None

        def key(self):
            return str(self.h)

        def status(self):
            st = self.h.status()
            if st == "ongoing":
                return ("ongoing", None)
            return ("finished", {1: 1, -1: 2, 0: None}[st])

        def legal_moves(self):
            if self._legal is None:
                self._legal = [nn for _, nn in self.h.legal_moves()]
            return list(self._legal)

        def moved(self, m):
            return Chess(self.h.moved(self.legal_moves().index(m)))

        def flipped(self):
            return Chess(self.h.flipped())

        @staticmethod
        def flip_move(m):  # mirror the ranks (chess/core.rs:40-53)
            nm = names[m]
            return index[nm[0] + str(9 - int(nm[1])) + nm[2] + str(9 - int(nm[3])) + nm[4:]]

        def planes(self):
            return [int(w) for w in self.h.planes().reshape(-1)]

    return Chess


# ----------------------------------------------------------------------------- value function


class NetValueFunction:
    """NNetwork::evaluate (net/mod.rs:74-103) around a raw ``net(planes_words) -> (logits, value)``."""

    def __init__(self, net, moves: int):
        self.net, self.moves, self.cache = net, moves, {}
        self.evals = 0

    def evaluate(self, pos):
        flipped = pos.turn != 1
        p = pos.flipped() if flipped else pos
        k = p.key()
        if k not in self.cache:
            logits, value = self.net(p.planes(), self.moves)
            self.evals += 1
            legal = p.legal_moves()
            probs = oracle.softmax_legal(logits, np.array(legal, dtype=np.uint32))  # same expf as the host library
            self.cache[k] = ([(m, F(pr)) for m, pr in zip(legal, probs)], F(value))
        probs, value = self.cache[k]
        if flipped:
            return [(type(pos).flip_move(m), pr) for m, pr in probs], F(-value)
        return list(probs), value


# --------------------------------------------------------------------------------------- MCTS


class Graph:
    def __init__(self):
        self.pos, self.out, self.edges = [], [], []

    def add_node(self, p):
        self.pos.append(p)
        self.out.append([])
        return len(self.pos) - 1

    def add_edge(self, a, b, m, prior):
        self.edges.append(dict(m=m, p=F(prior), n=0, w=F(0), src=a, dst=b))
        self.out[a].append(len(self.edges) - 1)
        return len(self.edges) - 1

    def edges_of(self, a):  # petgraph: newest first
        return reversed(self.out[a])


class MctsPlayer:
    def __init__(self, sim_num, explore_factor, value_func, temperature=0.0):
        self.sim_num, self.c, self.vf, self.temperature = sim_num, F(explore_factor), value_func, temperature
        self.g, self.root = Graph(), None

    def _heur(self, e, parent_n):
        exploit = F(0) if e["n"] == 0 else e["w"] / F(e["n"])
        explore = self.c * e["p"] * (np.sqrt(F(parent_n)) / F(1 + e["n"]))
        return exploit + explore

    def _select(self):
        path, node = [], self.root
        while True:
            if self.g.pos[node].status()[0] == "finished" or not self.g.out[node]:
                return path
            total = 1 + sum(self.g.edges[e]["n"] for e in self.g.out[node])
            best, vbest = None, None
            for e in self.g.edges_of(node):  # max_by: y replaces best unless cmp(best, y) == Greater
                v = self._heur(self.g.edges[e], total)
                if best is None or not (vbest > v):
                    best, vbest = e, v
            path.append(best)
            node = self.g.edges[best]["dst"]

    def _repetition(self, history, path):
        """detect_repetition (mcts/mod.rs:133-154): positions of the game history, then the targets of the
        path's edges; true as soon as one of them has been counted REPETITION_LIMIT times."""
        limit = getattr(type(history[-1]), "REPETITION_LIMIT", None)
        if not limit or limit <= 1:
            return False
        counts = {}
        for p in list(history) + [self.g.pos[self.g.edges[e]["dst"]] for e in path]:
            k = p.key()
            counts[k] = counts.get(k, 0) + 1
            if counts[k] >= limit:
                return True
        return False

    def _develop(self, history):
        assert self.sim_num > 1
        for _ in range(self.sim_num):
            path = self._select()
            leaf = self.root if not path else self.g.edges[path[-1]]["dst"]
            st = self.g.pos[leaf].status()
            if self._repetition(history, path):
                score = F(0)
            elif st[0] == "finished":
                score = F({1: 1.0, 2: -1.0, None: 0.0}[st[1]])
            else:
                per_move, score = self.vf.evaluate(self.g.pos[leaf])
                parent = self.g.pos[leaf]
                for m, p in per_move:
                    child = self.g.add_node(parent.moved(m))
                    self.g.add_edge(leaf, child, m, p)
            for e in path:
                edge = self.g.edges[e]
                edge["n"] += 1
                edge["w"] = F(edge["w"] + (score if self.g.pos[edge["src"]].turn == 1 else -score))

    def _find(self, pos, depth):
        layer = [self.root]
        for _ in range(depth):
            nxt = []
            for node in layer:
                if self.g.pos[node].key() == pos.key():
                    return node
                nxt += [self.g.edges[e]["dst"] for e in self.g.edges_of(node)]
            layer = nxt
        return None

    def _keep_subtree(self, sub):
        if sub == self.root:
            return
        ng = Graph()
        nroot = ng.add_node(self.g.pos[sub])
        stack = [(sub, nroot)]
        while stack:
            po, pn = stack.pop()
            for e in self.g.edges_of(po):
                ed = self.g.edges[e]
                cn = ng.add_node(self.g.pos[ed["dst"]])
                ne = ng.add_edge(pn, cn, ed["m"], ed["p"])
                ng.edges[ne]["n"], ng.edges[ne]["w"] = ed["n"], ed["w"]
                stack.append((ed["dst"], cn))
        self.g, self.root = ng, nroot

    def calc_moves_probabilities(self, history):
        pos = history[-1]
        if self.root is not None:
            node = self._find(pos, 3)
            if node is not None:
                self._keep_subtree(node)
            else:
                self.g, self.root = Graph(), None
        if self.root is None:
            self.root = self.g.add_node(pos)
        self._develop(history)
        visits = [(self.g.edges[e]["m"], self.g.edges[e]["n"]) for e in self.g.edges_of(self.root)]
        total = sum(n for _, n in visits)
        return visits, [(m, F(n) / F(total)) for m, n in visits]

    @staticmethod
    def choose_greedy(probs):
        best = None
        for m, p in probs:  # max_by(total_cmp): last maximum wins
            if best is None or not (best[1] > p):
                best = (m, p)
        return best[0]


def trace_game(game_cls, sim_num, explore_factor, max_plies=512, net=stub_net, forced=(), search_from=0):
    """One self-play game as a reference worker plays it (two persistent players, temperature 0).
    Returns [(chosen move, [(move, visits), ...]), ...] per searched ply, moves as nn indices.
    forced / search_from as cattus_sp_trace_game_ex: plies < len(forced) play forced[ply], plies <
    search_from are not searched.  The game ends on the position's status or, for games with a repetition
    limit, when a position has occurred that many times (Game::play_single_turn, chess/core.rs:438-450)."""
    vf = NetValueFunction(net, game_cls.MOVES)
    p1 = MctsPlayer(sim_num, explore_factor, vf)
    p2 = MctsPlayer(sim_num, explore_factor, vf)
    history, out, ply = [game_cls()], [], 0
    limit = getattr(game_cls, "REPETITION_LIMIT", None)
    seen, repetition = {history[0].key(): 1}, False
    while len(out) < max_plies and not repetition and history[-1].status()[0] == "ongoing":
        m = forced[ply] if ply < len(forced) else None
        if ply >= search_from:
            cur = p1 if history[-1].turn == 1 else p2
            visits, probs = cur.calc_moves_probabilities(history)
            chosen = MctsPlayer.choose_greedy(probs)
            out.append((chosen, visits))
            if m is None:
                m = chosen
        history.append(history[-1].moved(m))
        k = history[-1].key()
        seen[k] = seen.get(k, 0) + 1
        if limit and seen[k] >= limit:
            repetition = True
        ply += 1
    return out, vf.evals
