// dfpn.cc -- see dfpn.h
#include "dfpn.h"

#include <algorithm>

namespace nshogi {
namespace engine {
namespace shogi {

DfpnSolver::DfpnSolver(std::size_t TableEntriesLog2)
    : Table((std::size_t)1 << TableEntriesLog2), Mask(((uint64_t)1 << TableEntriesLog2) - 1), Levels(kMaxDepth + 2) {}

void DfpnSolver::look(uint64_t Key, uint32_t& Pn, uint32_t& Dn) const {
    const Entry& E = Table[Key & Mask];
    if (E.Gen == Gen && E.Key == Key) {
        Pn = E.Pn;
        Dn = E.Dn;
    } else {
        Pn = Dn = 1;
    }
}

void DfpnSolver::store(uint64_t Key, uint32_t Pn, uint32_t Dn) {
    Entry& E = Table[Key & Mask];
    // keep a finished (proved / disproved) entry of another position rather than overwrite it with an open one
    if (E.Gen == Gen && E.Key != Key && (E.Pn == 0 || E.Dn == 0) && Pn != 0 && Dn != 0) return;
    E.Key = Key;
    E.Pn = Pn;
    E.Dn = Dn;
    E.Gen = Gen;
}

// Children of a node with the hash of the position each leads to.
void DfpnSolver::expand(State& S, bool Or, Level& L) const {
    L.Moves.clear();
    L.Keys.clear();
    MoveList All;
    S.generateLegalMoves(All);
    for (const Move& M : All) {
        if (Or && !S.givesCheck(M)) continue;
        L.Moves.push_back(M);
    }
    for (const Move& M : L.Moves) {
        S.doMove(M);
        L.Keys.push_back(S.hash());
        S.undoMove();
    }
}

// Returns true if the result is a failure of the attacker (Dn == 0) that leaned on a repetition
// or on the depth cap: valid on this path only, so it is not stored as final.
bool DfpnSolver::search(State& S, uint32_t ThPn, uint32_t ThDn, bool Or, int Depth, uint32_t& Pn, uint32_t& Dn) {
    ++Nodes;
    const uint64_t Key = S.hash();
    Level& L = Levels[Depth];
    expand(S, Or, L);
    if (L.Moves.empty()) {
        // attacker without a check: failed; defender without a reply: mated
        Pn = Or ? kInf : 0;
        Dn = Or ? 0 : kInf;
        store(Key, Pn, Dn);
        return false;
    }
    if (Depth >= kMaxDepth) {
        Pn = kInf;
        Dn = 0;
        return true;
    }
    Path.push_back(Key);
    const std::size_t N = L.Moves.size();
    L.Pn.resize(N);
    L.Dn.resize(N);
    L.Dep.assign(N, 0);
    for (std::size_t I = 0; I < N; ++I) {
        if (std::find(Path.begin(), Path.end(), L.Keys[I]) != Path.end()) {
            L.Pn[I] = kInf; // a repetition inside the search: the attacker gets nowhere
            L.Dn[I] = 0;
            L.Dep[I] = 1;
        } else {
            look(L.Keys[I], L.Pn[I], L.Dn[I]);
        }
    }
    bool Dependent = false;
    for (;;) {
        uint64_t SumPn = 0, SumDn = 0;
        uint32_t MinPn = kInf, MinDn = kInf, First = kInf + 1, Second = kInf + 1;
        std::size_t Best = 0;
        Dependent = false;
        for (std::size_t I = 0; I < N; ++I) {
            SumPn += L.Pn[I];
            SumDn += L.Dn[I];
            MinPn = std::min(MinPn, L.Pn[I]);
            MinDn = std::min(MinDn, L.Dn[I]);
            if (L.Dn[I] == 0 && L.Dep[I]) Dependent = true;
            const uint32_t Sel = Or ? L.Pn[I] : L.Dn[I]; // descend into the child minimising this
            if (Sel < First) {
                Second = First;
                First = Sel;
                Best = I;
            } else if (Sel < Second) {
                Second = Sel;
            }
        }
        if (Or) {
            Pn = MinPn;
            Dn = (uint32_t)std::min<uint64_t>(SumDn, kInf);
        } else {
            Pn = (uint32_t)std::min<uint64_t>(SumPn, kInf);
            Dn = MinDn;
        }
        if (Pn == 0) Dn = kInf;
        if (Dn == 0) Pn = kInf;
        if (Pn >= ThPn || Dn >= ThDn || Nodes >= MaxNodes) break;
        uint32_t NextPn, NextDn;
        if (Or) {
            NextPn = (uint32_t)std::min<uint64_t>(ThPn, (uint64_t)Second + 1);
            NextDn = (uint32_t)std::min<uint64_t>((uint64_t)ThDn - Dn + L.Dn[Best], kInf);
        } else {
            NextDn = (uint32_t)std::min<uint64_t>(ThDn, (uint64_t)Second + 1);
            NextPn = (uint32_t)std::min<uint64_t>((uint64_t)ThPn - Pn + L.Pn[Best], kInf);
        }
        S.doMove(L.Moves[Best]);
        uint32_t RPn, RDn;
        const bool Dep = search(S, NextPn, NextDn, !Or, Depth + 1, RPn, RDn);
        S.undoMove();
        L.Pn[Best] = RPn;
        L.Dn[Best] = RDn;
        L.Dep[Best] = Dep ? 1 : 0;
    }
    Path.pop_back();
    if (Dn == 0 && Dependent) return true;
    store(Key, Pn, Dn);
    return false;
}

// Follows proven table entries below the proven root move (the PV ends early if entries were lost).
void DfpnSolver::extractPv(State& S, Move RootMove) {
    Pv.assign(1, RootMove);
    S.doMove(RootMove);
    int Made = 1;
    bool Or = false;
    for (int Depth = 1; Depth < kMaxDepth; ++Depth, Or = !Or) {
        Level& L = Levels[0];
        expand(S, Or, L);
        std::size_t Pick = L.Moves.size();
        for (std::size_t I = 0; I < L.Moves.size(); ++I) {
            uint32_t CPn, CDn;
            look(L.Keys[I], CPn, CDn);
            if (CPn == 0) {
                Pick = I;
                break;
            }
        }
        if (Pick == L.Moves.size()) break;
        Pv.push_back(L.Moves[Pick]);
        S.doMove(L.Moves[Pick]);
        ++Made;
    }
    while (Made-- > 0) S.undoMove();
}

Move DfpnSolver::solve(State& S, uint64_t NodeBudget) {
    ++Gen;
    if (Gen == 0) { // generation counter wrapped: really clear
        std::fill(Table.begin(), Table.end(), Entry());
        Gen = 1;
    }
    Nodes = 0;
    MaxNodes = NodeBudget;
    Path.clear();
    Pv.clear();
    if (NodeBudget == 0) return Move();
    uint32_t Pn, Dn;
    search(S, kInf, kInf, true, 0, Pn, Dn);
    if (Pn != 0) return Move();
    const Level& Root = Levels[0];
    Move RootMove;
    for (std::size_t I = 0; I < Root.Moves.size(); ++I) {
        if (Root.Pn[I] == 0) {
            RootMove = Root.Moves[I];
            break;
        }
    }
    if (!RootMove.isNone()) extractPv(S, RootMove);
    return RootMove;
}

} // namespace shogi
} // namespace engine
} // namespace nshogi
