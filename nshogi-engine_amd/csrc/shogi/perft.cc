// perft.cc -- validation tool of the shogi core.
//   perft <depth> [sfen]            node counts (public reference values pin movegen)
//   movecount <sfen>                number of legal moves
//   crosscheck <games> <seed>       random playouts: fast generator == slow generator,
//                                   do/undo restores the hash, sfen round-trips
#include "shogi.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
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
