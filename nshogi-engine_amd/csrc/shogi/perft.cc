// perft.cc -- validation tool of the shogi core.
//   perft <depth> [sfen]            node counts (public reference values pin movegen)
//   movecount <sfen>                number of legal moves
//   crosscheck <games> <seed>       random playouts: fast generator == slow generator,
//                                   do/undo restores the hash, sfen round-trips
//   features <games> <seed> <maxply> <blackdraw> [stop]
//                                   random playouts (Moves[Mt() % size], the shape of
//                                   src/test/test_extractbit.cc:66-91); one line per position:
//                                   sfen TAB hex(86 x 16-byte feature bitboards) TAB usi:index ...
//   featuresat <maxply> <blackdraw> <sfen>   the same line for one given position
#include "shogi.h"
#include "dfpn.h"
#include "features.h"

#include <functional>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <chrono>
#include <random>
#include <string>

using namespace nshogi::engine::shogi;

static std::string joinArgs(int Argc, char** Argv, int From) {
    std::string S;
    for (int I = From; I < Argc; ++I) S += (I > From ? " " : "") + std::string(Argv[I]);
    return S;
}

int main(int Argc, char** Argv) {
    if (Argc < 2) return 2;
    const std::string Cmd = Argv[1];
    if (Cmd == "perft") {
        const int Depth = std::atoi(Argv[2]);
        State S = Argc > 3 ? State::fromSfen(joinArgs(Argc, Argv, 3)) : State();
        for (int D = 1; D <= Depth; ++D) std::cout << D << " " << S.perft(D) << std::endl;
        return 0;
    }
    if (Cmd == "movecount") {
        State S = State::fromSfen(joinArgs(Argc, Argv, 2));
        MoveList L;
        S.generateLegalMoves(L);
        MoveList L2;
        S.generateLegalMovesSlow(L2);
        std::cout << L.size() << " " << L2.size() << " incheck " << S.inCheck() << " declare " << S.canDeclare() << std::endl;
        return 0;
    }
    if (Cmd == "mate") { // mate <depth> <sfen>: prints the mating move (or "none") with and without the prefilter
        const int Depth = std::atoi(Argv[2]);
        State S = State::fromSfen(joinArgs(Argc, Argv, 3));
        const Move A = S.findMate(Depth, true), B = S.findMate(Depth, false);
        std::cout << (A.isNone() ? std::string("none") : moveToUsi(A)) << " "
                  << (B.isNone() ? std::string("none") : moveToUsi(B)) << std::endl;
        return 0;
    }
    if (Cmd == "matecheck") { // matecheck <games> <seed>: prefiltered vs unfiltered search on random playouts
        const int Games = std::atoi(Argv[2]);
        std::mt19937_64 Rng((uint64_t)std::atoll(Argv[3]));
        uint64_t Positions = 0, Mate1 = 0, Mate3 = 0;
        for (int G = 0; G < Games; ++G) {
            State S;
            for (int Ply = 0; Ply < 300; ++Ply) {
                MoveList L;
                S.generateLegalMoves(L);
                if (S.hasLegalMove() != (L.size() > 0)) { std::cout << "hasLegalMove mismatch at " << S.toSfen() << std::endl; return 1; }
                if (L.size() == 0 || S.repetitionStatus(true) != NoRepetition) break;
                const uint64_t H = S.hash();
                State::CheckInfo CI;
                S.checkInfo(CI);
                for (const Move& M : L) { // givesCheck vs make-the-move-and-look, and the per-position table vs both
                    const bool Quick = S.givesCheck(M);
                    if (S.givesCheck(M, CI) != Quick) { std::cout << "check table mismatch at " << S.toSfen() << " move " << moveToUsi(M) << std::endl; return 1; }
                    S.doMove(M);
                    const bool Slow = S.inCheck();
                    S.undoMove();
                    if (Quick != Slow) { std::cout << "givesCheck mismatch at " << S.toSfen() << " move " << moveToUsi(M) << " quick " << Quick << std::endl; return 1; }
                }
                const Move A = S.findMate(3, true), B = S.findMate(3, false);
                if (S.hash() != H) { std::cout << "findMate changed the position " << S.toSfen() << std::endl; return 1; }
                if (A.isNone() != B.isNone()) { std::cout << "MISMATCH at " << S.toSfen() << std::endl; return 1; }
                ++Positions;
                if (!A.isNone()) {
                    ++Mate3;
                    if (!S.findMate(1, true).isNone()) ++Mate1;
                    // verify: after A the defender is in check and every reply allows a mate in one
                    S.doMove(A);
                    if (!S.inCheck()) { std::cout << "mating move gives no check " << S.toSfen() << std::endl; return 1; }
                    MoveList R;
                    S.generateLegalMoves(R);
                    for (const Move& Rm : R) {
                        S.doMove(Rm);
                        if (S.findMate(1, false).isNone()) { std::cout << "reply escapes " << S.toSfen() << std::endl; return 1; }
                        S.undoMove();
                    }
                    S.undoMove();
                }
                S.doMove(L[(int)(Rng() % (uint64_t)L.size())]);
            }
        }
        std::cout << "positions " << Positions << " mate3 " << Mate3 << " mate1 " << Mate1 << " ok" << std::endl;
        return 0;
    }
    if (Cmd == "matesample") { // matesample <seed>: a random-playout position with a mate in three but not in one
        std::mt19937_64 Rng((uint64_t)std::atoll(Argv[2]));
        for (int G = 0; G < 1000; ++G) {
            State S;
            for (int Ply = 0; Ply < 200; ++Ply) {
                MoveList L;
                S.generateLegalMoves(L);
                if (L.size() == 0 || S.repetitionStatus(true) != NoRepetition) break;
                if (S.findMate(1).isNone() && !S.findMate(3).isNone()) {
                    std::cout << S.toSfen() << std::endl;
                    return 0;
                }
                S.doMove(L[(int)(Rng() % (uint64_t)L.size())]);
            }
        }
        return 1;
    }
    if (Cmd == "matebench") { // matebench <games> <seed>: ns per findMate(3) and per generateLegalMoves on random playouts
        const int Games = std::atoi(Argv[2]);
        std::mt19937_64 Rng((uint64_t)std::atoll(Argv[3]));
        std::vector<State> Pos;
        for (int G = 0; G < Games; ++G) {
            State S;
            for (int Ply = 0; Ply < 200; ++Ply) {
                MoveList L;
                S.generateLegalMoves(L);
                if (L.size() == 0 || S.repetitionStatus(true) != NoRepetition) break;
                Pos.push_back(S);
                S.doMove(L[(int)(Rng() % (uint64_t)L.size())]);
            }
        }
        auto T0 = std::chrono::steady_clock::now();
        uint64_t Found = 0, Total = 0;
        for (State& S : Pos) Found += !S.findMate(3).isNone();
        auto T1 = std::chrono::steady_clock::now();
        for (State& S : Pos) { MoveList L; S.generateLegalMoves(L); Total += L.size(); }
        auto T2 = std::chrono::steady_clock::now();
        std::cout << "positions " << Pos.size() << " mates " << Found << " findMate(3) "
                  << std::chrono::duration<double, std::nano>(T1 - T0).count() / Pos.size() << " ns, generateLegalMoves "
                  << std::chrono::duration<double, std::nano>(T2 - T1).count() / Pos.size() << " ns (" << (double)Total / Pos.size() << " moves)" << std::endl;
        return 0;
    }
    if (Cmd == "dfpn") { // dfpn <nodes> <sfen>: mating move (or "none"), nodes used, principal variation
        State S = State::fromSfen(joinArgs(Argc, Argv, 3));
        DfpnSolver Solver;
        const Move M = Solver.solve(S, (uint64_t)std::atoll(Argv[2]));
        std::cout << (M.isNone() ? std::string("none") : moveToUsi(M)) << " nodes " << Solver.nodes() << " pv";
        for (const Move& P : Solver.pv()) std::cout << " " << moveToUsi(P);
        std::cout << std::endl;
        return 0;
    }
    if (Cmd == "dfpncheck") { // dfpncheck <games> <seed> <nodes>: soundness and shallow completeness on random playouts
        const int Games = std::atoi(Argv[2]);
        std::mt19937_64 Rng((uint64_t)std::atoll(Argv[3]));
        const uint64_t Budget = (uint64_t)std::atoll(Argv[4]);
        DfpnSolver Solver, Sub;
        uint64_t Positions = 0, Mate3 = 0, Found = 0, Deeper = 0, Verified = 0, GaveUp = 0, NodesSum = 0, MaxPv = 0;
        // Every defence must again lose to a proven move, down to positions without a reply.
        // Returns 1 verified, 0 refuted, -1 out of verification budget.
        uint64_t Work = 0;
        std::function<int(State&, Move, int)> Verify = [&](State& S, Move A, int Depth) -> int {
            if (++Work > 20000 || Depth > 60) return -1;
            S.doMove(A);
            int Result = 1;
            if (!S.inCheck()) Result = 0;
            MoveList R;
            S.generateLegalMoves(R);
            for (int I = 0; I < R.size() && Result == 1; ++I) {
                S.doMove(R[I]);
                const Move Next = Sub.solve(S, Budget);
                if (Next.isNone()) Result = Sub.nodes() >= Budget ? -1 : 0;
                else Result = Verify(S, Next, Depth + 2);
                S.undoMove();
            }
            S.undoMove();
            return Result;
        };
        for (int G = 0; G < Games; ++G) {
            State S;
            for (int Ply = 0; Ply < 300; ++Ply) {
                MoveList L;
                S.generateLegalMoves(L);
                if (L.size() == 0 || S.repetitionStatus(true) != NoRepetition) break;
                ++Positions;
                const uint64_t H = S.hash();
                const bool Shallow = !S.findMate(3).isNone();
                const Move A = Solver.solve(S, Budget);
                NodesSum += Solver.nodes();
                if (S.hash() != H) { std::cout << "solve did not restore the position " << S.toSfen() << std::endl; return 1; }
                Mate3 += Shallow;
                if (Shallow && A.isNone() && Solver.nodes() < Budget) {
                    std::cout << "dfpn disproved a position with a mate in three " << S.toSfen() << std::endl;
                    return 1;
                }
                if (!A.isNone()) {
                    ++Found;
                    if (!Shallow) ++Deeper;
                    MaxPv = std::max<uint64_t>(MaxPv, Solver.pv().size());
                    Work = 0;
                    const int V = Verify(S, A, 1);
                    if (V == 0) { std::cout << "unsound mate " << moveToUsi(A) << " at " << S.toSfen() << std::endl; return 1; }
                    if (V == 1) ++Verified; else ++GaveUp;
                }
                S.doMove(L[(int)(Rng() % (uint64_t)L.size())]);
            }
        }
        std::cout << "positions " << Positions << " mate3 " << Mate3 << " dfpn " << Found << " deeper " << Deeper
                  << " verified " << Verified << " gaveup " << GaveUp << " maxpv " << MaxPv << " avg_nodes "
                  << (double)NodesSum / (double)std::max<uint64_t>(Positions, 1) << " ok" << std::endl;
        return 0;
    }
    if (Cmd == "dfpnbench") { // dfpnbench <games> <seed> <nodes>: ns per node expansion and per solve on random playouts
        const int Games = std::atoi(Argv[2]);
        std::mt19937_64 Rng((uint64_t)std::atoll(Argv[3]));
        const uint64_t Budget = (uint64_t)std::atoll(Argv[4]);
        std::vector<State> Pos;
        for (int G = 0; G < Games; ++G) {
            State S;
            for (int Ply = 0; Ply < 200; ++Ply) {
                MoveList L;
                S.generateLegalMoves(L);
                if (L.size() == 0 || S.repetitionStatus(true) != NoRepetition) break;
                Pos.push_back(S);
                S.doMove(L[(int)(Rng() % (uint64_t)L.size())]);
            }
        }
        DfpnSolver Solver;
        uint64_t Nodes = 0, Found = 0, Exhausted = 0;
        auto T0 = std::chrono::steady_clock::now();
        for (State& S : Pos) {
            Found += !Solver.solve(S, Budget).isNone();
            Nodes += Solver.nodes();
            Exhausted += Solver.nodes() >= Budget;
        }
        const double Ns = std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - T0).count();
        std::cout << "positions " << Pos.size() << " mates " << Found << " budget_exhausted " << Exhausted << " nodes/solve "
                  << (double)Nodes / Pos.size() << " ns/node " << Ns / Nodes << " us/solve " << Ns / 1000.0 / Pos.size() << std::endl;
        return 0;
    }
    if (Cmd == "features" || Cmd == "featuresat") {
        // Dumps what the self-play path feeds the evaluator (FeatureType::constructAt,
        // selfplay/evaluationworker.cc:87-92) and the policy index of every legal move
        // (ml::getMoveIndex, selfplay/frame.cc:102-105); the checks live in tests/shogi_ref.py,
        // which rebuilds both from the SFEN text alone.
        auto Dump = [](const State& S, const StateConfig& Config) {
            FeaturePlane Planes[NumFeaturePlanes];
            buildFeatures(S, Config, Planes);
            std::string Hex;
            static const char* Digits = "0123456789abcdef";
            const unsigned char* Bytes = reinterpret_cast<const unsigned char*>(Planes);
            for (size_t I = 0; I < sizeof(Planes); ++I) {
                Hex += Digits[Bytes[I] >> 4];
                Hex += Digits[Bytes[I] & 15];
            }
            std::cout << S.toSfen() << "\t" << Hex << "\t";
            MoveList L;
            S.generateLegalMoves(L);
            for (int I = 0; I < L.size(); ++I)
                std::cout << (I ? " " : "") << moveToUsi(L[I]) << ":" << moveIndex(S.sideToMove(), L[I]);
            std::cout << "\n";
        };
        StateConfig Config;
        if (Cmd == "featuresat") {
            Config.MaxPly = (uint16_t)std::atoi(Argv[2]);
            Config.BlackDrawValue = (float)std::atof(Argv[3]);
            Config.WhiteDrawValue = 1.0f - Config.BlackDrawValue;
            Dump(State::fromSfen(joinArgs(Argc, Argv, 4)), Config);
            return 0;
        }
        const int Games = std::atoi(Argv[2]);
        std::mt19937_64 Rng((uint64_t)std::atoll(Argv[3]));
        Config.MaxPly = (uint16_t)std::atoi(Argv[4]);
        Config.BlackDrawValue = (float)std::atof(Argv[5]);
        Config.WhiteDrawValue = 1.0f - Config.BlackDrawValue;
        const int Stop = Argc > 6 ? std::atoi(Argv[6]) : (int)Config.MaxPly;
        for (int G = 0; G < Games; ++G) {
            State S;
            Dump(S, Config);
            for (int Ply = 0; Ply < Stop; ++Ply) {
                MoveList L;
                S.generateLegalMoves(L);
                if (L.size() == 0) break;
                S.doMove(L[(int)(Rng() % (uint64_t)L.size())]);
                Dump(S, Config);
            }
        }
        return 0;
    }
    if (Cmd == "moves") {
        State S = State::fromSfen(joinArgs(Argc, Argv, 2));
        MoveList L;
        S.generateLegalMoves(L);
        for (const Move& M : L) std::cout << moveToUsi(M) << " ";
        std::cout << std::endl;
        return 0;
    }
    if (Cmd == "crosscheck") {
        const int Games = std::atoi(Argv[2]);
        std::mt19937_64 Rng((uint64_t)std::atoll(Argv[3]));
        uint64_t Positions = 0, MaxMoves = 0, Mates = 0, Repetitions = 0, Declares = 0;
        for (int G = 0; G < Games; ++G) {
            State S;
            for (int Ply = 0; Ply < 400; ++Ply) {
                MoveList A, B;
                S.generateLegalMoves(A);
                S.generateLegalMovesSlow(B);
                ++Positions;
                std::vector<uint32_t> VA, VB;
                for (const Move& M : A) VA.push_back(M.V);
                for (const Move& M : B) VB.push_back(M.V);
                std::sort(VA.begin(), VA.end());
                std::sort(VB.begin(), VB.end());
                if (VA != VB) {
                    std::cout << "MISMATCH at " << S.toSfen() << " fast " << VA.size() << " slow " << VB.size() << std::endl;
                    for (uint32_t V : VA) if (!std::binary_search(VB.begin(), VB.end(), V)) { Move M; M.V = V; std::cout << " fast-only " << moveToUsi(M) << std::endl; }
                    for (uint32_t V : VB) if (!std::binary_search(VA.begin(), VA.end(), V)) { Move M; M.V = V; std::cout << " slow-only " << moveToUsi(M) << std::endl; }
                    return 1;
                }
                MaxMoves = std::max<uint64_t>(MaxMoves, A.size());
                if (A.size() == 0) { ++Mates; break; }
                if (S.repetitionStatus(true) != NoRepetition) { ++Repetitions; break; }
                if (S.canDeclare()) { ++Declares; }
                // sfen round trip and do/undo hash invariance
                const std::string Sf = S.toSfen();
                State T = State::fromSfen(Sf);
                if (T.hash() != S.hash() || T.toSfen() != Sf) { std::cout << "SFEN roundtrip failed " << Sf << std::endl; return 1; }
                const uint64_t H = S.hash();
                const Move M = A[(int)(Rng() % (uint64_t)A.size())];
                if (S.moveFrom16(M.move16()).V != M.V) { std::cout << "move16 rebuild failed" << std::endl; return 1; }
                S.doMove(M);
                S.undoMove();
                if (S.hash() != H || S.toSfen() != Sf) { std::cout << "undo failed " << Sf << " " << moveToUsi(M) << std::endl; return 1; }
                S.doMove(M);
            }
        }
        std::cout << "ok positions " << Positions << " max_moves " << MaxMoves << " mates " << Mates
                  << " repetitions " << Repetitions << " declares " << Declares << std::endl;
        return 0;
    }
    return 2;
}
