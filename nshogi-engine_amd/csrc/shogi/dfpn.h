// dfpn.h -- depth-first proof-number search for forced mates by consecutive checks.
//
// Role of libnshogi's solver::dfpn::Solver as the reference calls it once per move in
// selfplay::Worker::judge (/root/reference/src/selfplay/worker.cc:516-524:
// `Solver.solve(State, 100000, 0)` -> a mating first move or none).  libnshogi is absent, so this
// is the published algorithm (Nagai 2002: df-pn with proof / disproof numbers kept in a
// transposition table and the 1+epsilon-free second-best thresholds), written against this
// build's rules core.  Attacker (OR) nodes try only legal moves that give check, defender (AND)
// nodes every legal reply; a defender without a reply is mated (a mate by a dropped pawn never
// appears: the generator does not emit it).  A position repeated on the search path or deeper than
// kMaxDepth counts as a failure of the attacker, and such path-dependent failures are not stored.
// Soundness -- a returned move really forces mate -- is tested by replaying the proof against every
// defence (perft "dfpncheck"); completeness is bounded by the node budget.
#ifndef NSG_SHOGI_DFPN_H
#define NSG_SHOGI_DFPN_H

#include "shogi.h"

#include <cstdint>
#include <vector>

namespace nshogi {
namespace engine {
namespace shogi {

class DfpnSolver {
 public:
    explicit DfpnSolver(std::size_t TableEntriesLog2 = 16);

    // A move of the side to move that forces checkmate by consecutive checks, or a none move if
    // none was proved within MaxNodes node expansions.  S is restored before returning.
    Move solve(State& S, uint64_t MaxNodes);
    uint64_t nodes() const { return Nodes; }          // expansions of the last solve()
    // Principal variation of the last successful solve(): attacker's proven move, then for every
    // ply the first proven continuation (the defender's longest-surviving reply is not searched for).
    const std::vector<Move>& pv() const { return Pv; }

    static constexpr int kMaxDepth = 96;

 private:
    static constexpr uint32_t kInf = 1u << 30;
    struct Entry {
        uint64_t Key = 0;
        uint32_t Pn = 1, Dn = 1;
        uint32_t Gen = 0;
    };
    struct Level {
        std::vector<Move> Moves;
        std::vector<uint64_t> Keys;
        std::vector<uint32_t> Pn, Dn; // the children's numbers as this node last saw them
        std::vector<uint8_t> Dep;     // child's failure is valid on this path only
    };
    bool search(State& S, uint32_t ThPn, uint32_t ThDn, bool Or, int Depth, uint32_t& Pn, uint32_t& Dn);
    void look(uint64_t Key, uint32_t& Pn, uint32_t& Dn) const;
    void store(uint64_t Key, uint32_t Pn, uint32_t Dn);
    void expand(State& S, bool Or, Level& L) const;
    void extractPv(State& S, Move RootMove);

    std::vector<Entry> Table;
    uint64_t Mask;
    uint32_t Gen = 0;
    uint64_t Nodes = 0, MaxNodes = 0;
    std::vector<Level> Levels;
    std::vector<uint64_t> Path;
    std::vector<Move> Pv;
};

} // namespace shogi
} // namespace engine
} // namespace nshogi

#endif
