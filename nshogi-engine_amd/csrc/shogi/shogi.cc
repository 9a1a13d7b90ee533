// shogi.cc -- rules core (see shogi.h).
#include "shogi.h"

#include <cassert>
#include <cstring>
#include <sstream>

namespace nshogi {
namespace engine {
namespace shogi {

namespace {

// direction i = (dfile, drank); i+4 (mod 8) is the opposite direction.
constexpr int kDF[8] = {0, -1, -1, -1, 0, 1, 1, 1};
constexpr int kDR[8] = {-1, -1, 0, 1, 1, 1, 0, -1};
enum : uint8_t { N = 1, NE = 2, E = 4, SE = 8, S = 16, SW = 32, W = 64, NW = 128 };

// Black's movement masks by piece type (White's are rotated by 4 directions).
constexpr uint8_t kStep[PT_Num] = {
    0, N, 0, 0, (uint8_t)(N | NE | NW | SE | SW), 0, 0, (uint8_t)(N | NE | NW | E | W | S), 0xff,
    (uint8_t)(N | NE | NW | E | W | S), (uint8_t)(N | NE | NW | E | W | S), (uint8_t)(N | NE | NW | E | W | S),
    (uint8_t)(N | NE | NW | E | W | S), (uint8_t)(N | E | S | W), (uint8_t)(NE | SE | SW | NW)};
constexpr uint8_t kSlide[PT_Num] = {
    0, 0, N, 0, 0, (uint8_t)(NE | SE | SW | NW), (uint8_t)(N | E | S | W), 0, 0,
    0, 0, 0, 0, (uint8_t)(NE | SE | SW | NW), (uint8_t)(N | E | S | W)};

inline uint8_t rot(uint8_t M, Color C) { return C == Black ? M : (uint8_t)((M << 4) | (M >> 4)); }
inline uint8_t stepMask(Piece P) { return rot(kStep[typeOf(P)], colorOf(P)); }
inline uint8_t slideMask(Piece P) { return rot(kSlide[typeOf(P)], colorOf(P)); }

inline bool onBoard(int F, int R) { return (unsigned)F < 9u && (unsigned)R < 9u; }
inline int forward(Color C) { return C == Black ? -1 : 1; }
inline bool inZone(Color C, int Rank) { return C == Black ? Rank <= 2 : Rank >= 6; }
// ranks a piece of this type may not stand on unpromoted (relative rank 0 = last rank)
inline int relRank(Color C, int Rank) { return C == Black ? Rank : 8 - Rank; }

uint64_t splitmix(uint64_t& X) {
    X += 0x9e3779b97f4a7c15ULL;
    uint64_t Z = X;
    Z = (Z ^ (Z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    Z = (Z ^ (Z >> 27)) * 0x94d049bb133111ebULL;
    return Z ^ (Z >> 31);
}

struct Zobrist {
    uint64_t PieceSq[32][NumSquares];
    uint64_t Hand[2][8][20];
    uint64_t Side;
    Zobrist() {
        uint64_t S = 20240203;
        for (auto& A : PieceSq)
            for (auto& V : A) V = splitmix(S);
        for (auto& A : Hand)
            for (auto& B : A)
                for (auto& V : B) V = splitmix(S);
        Side = splitmix(S);
    }
};
const Zobrist& zob() {
    static const Zobrist Z;
    return Z;
}

const char* kSfenPiece = " PLNSBRGK";

} // namespace

// (set once, before main: every State reads it, none writes it -- it used to be assigned in fromSfen, a
// benign but real data race once several host threads parsed positions)
uint64_t State::SideKey = zob().Side;

void State::put(int Sq, Piece P) {
    Board[Sq] = P;
    BoardHash ^= zob().PieceSq[P][Sq];
    if (typeOf(P) == King) KingSq[colorOf(P)] = Sq;
}

void State::remove(int Sq) {
    BoardHash ^= zob().PieceSq[Board[Sq]][Sq];
    Board[Sq] = 0;
}

State::State() : State(fromSfen("lnsgkgsnl/1r5b1/ppppppppp/9/9/9/PPPPPPPPP/1B5R1/LNSGKGSNL b - 1")) {
}

State State::fromSfen(const std::string& Sfen) {
    State S(0);
    std::istringstream In(Sfen);
    std::string BoardStr, SideStr, HandStr;
    int Ply = 1;
    In >> BoardStr >> SideStr >> HandStr >> Ply;
    int Col = 0, Rank = 0;
    bool Promo = false;
    for (char C : BoardStr) {
        if (C == '/') {
            ++Rank;
            Col = 0;
        } else if (C == '+') {
            Promo = true;
        } else if (C >= '1' && C <= '9') {
            Col += C - '0';
        } else {
            const Color Cl = (C >= 'a' && C <= 'z') ? White : Black;
            const char U = (char)(Cl == White ? C - 'a' + 'A' : C);
            const char* P = std::strchr(kSfenPiece, U);
            PieceType T = (PieceType)(P - kSfenPiece);
            if (Promo) T = promote(T);
            Promo = false;
            S.put(makeSquare(8 - Col, Rank), makePiece(Cl, T));
            ++Col;
        }
    }
    S.Side = SideStr == "w" ? White : Black;
    if (HandStr != "-") {
        int Count = 0;
        for (char C : HandStr) {
            if (C >= '0' && C <= '9') {
                Count = Count * 10 + (C - '0');
                continue;
            }
            const Color Cl = (C >= 'a' && C <= 'z') ? White : Black;
            const char U = (char)(Cl == White ? C - 'a' + 'A' : C);
            const PieceType T = (PieceType)(std::strchr(kSfenPiece, U) - kSfenPiece);
            const int N2 = Count == 0 ? 1 : Count;
            for (int I = 0; I < N2; ++I) {
                S.HandHash ^= zob().Hand[Cl][T][S.Hands[Cl][T]];
                ++S.Hands[Cl][T];
            }
            Count = 0;
        }
    }
    S.PlyOffset = Ply - 1;
    return S;
}

// private "empty board" constructor used by fromSfen
State::State(int) : Side(Black), PlyOffset(0), BoardHash(0), HandHash(0) {
    std::memset(Board, 0, sizeof(Board));
    std::memset(Hands, 0, sizeof(Hands));
    KingSq[0] = KingSq[1] = -1;
}

std::string State::toSfen() const {
    std::string Out;
    for (int Rank = 0; Rank < 9; ++Rank) {
        int Empty = 0;
        for (int Col = 0; Col < 9; ++Col) {
            const Piece P = Board[makeSquare(8 - Col, Rank)];
            if (!P) {
                ++Empty;
                continue;
            }
            if (Empty) Out += (char)('0' + Empty);
            Empty = 0;
            if (isPromoted(typeOf(P))) Out += '+';
            const char C = kSfenPiece[demote(typeOf(P))];
            Out += colorOf(P) == White ? (char)(C - 'A' + 'a') : C;
        }
        if (Empty) Out += (char)('0' + Empty);
        if (Rank != 8) Out += '/';
    }
    Out += Side == Black ? " b " : " w ";
    std::string H;
    const PieceType Order[7] = {Rook, Bishop, Gold, Silver, Knight, Lance, Pawn};
    for (int C = 0; C < 2; ++C)
        for (PieceType T : Order) {
            const int N2 = Hands[C][T];
            if (!N2) continue;
            if (N2 > 1) H += std::to_string(N2);
            const char Ch = kSfenPiece[T];
            H += C == White ? (char)(Ch - 'A' + 'a') : Ch;
        }
    Out += H.empty() ? "-" : H;
    Out += " " + std::to_string(ply() + 1);
    return Out;
}

// Is `Sq` attacked by a piece of colour `By`, with `RemovedSq` treated as empty and
// `AddedSq` treated as an (opaque) blocker?  Early exit.
bool State::attackedWithout(int Sq, Color By, int RemovedSq, int AddedSq) const {
    const int F0 = fileOf(Sq), R0 = rankOf(Sq);
    for (int D = 0; D < 8; ++D) {
        int F = F0 + kDF[D], R = R0 + kDR[D];
        const uint8_t Back = (uint8_t)(1u << ((D + 4) & 7)); // direction from the attacker toward Sq
        bool First = true;
        while (onBoard(F, R)) {
            const int S2 = makeSquare(F, R);
            Piece P = (S2 == RemovedSq) ? (Piece)0 : Board[S2];
            if (S2 == AddedSq) break; // blocker that does not attack along this line
            if (P) {
                if (colorOf(P) == By) {
                    if (slideMask(P) & Back) return true;
                    if (First && (stepMask(P) & Back)) return true;
                }
                break;
            }
            First = false;
            F += kDF[D];
            R += kDR[D];
        }
    }
    // knights: a By knight on (F0 -+ 1, R0 - 2*forward(By)) jumps to Sq
    const int KR = R0 - 2 * forward(By);
    for (int DF = -1; DF <= 1; DF += 2) {
        if (onBoard(F0 + DF, KR)) {
            const int S2 = makeSquare(F0 + DF, KR);
            if (S2 != RemovedSq && Board[S2] == makePiece(By, Knight)) return true;
        }
    }
    return false;
}

// ---------------------------------------------------------------------------
// move generation
// ---------------------------------------------------------------------------
namespace {
struct GenCtx {
    bool Target[NumSquares]; // squares a non-king move must land on (check evasion); all true when not in check
    int8_t PinDir[NumSquares]; // -1 or the direction index of the pin line
};

inline void pushMoves(MoveList& Out, Color Us, int From, int To, PieceType T, PieceType Cap) {
    const int RelTo = relRank(Us, rankOf(To));
    if (canPromoteType(T) && (inZone(Us, rankOf(From)) || inZone(Us, rankOf(To)))) {
        Out.push(Move::make(From, To, true, T, Cap));
    }
    // unpromoted: not where the piece could never move again
    if ((T == Pawn || T == Lance) && RelTo == 0) return;
    if (T == Knight && RelTo <= 1) return;
    Out.push(Move::make(From, To, false, T, Cap));
}
} // namespace

void State::generateLegalMoves(MoveList& Out) const {
    genLegal<false>(Out);
}

bool State::hasLegalMove() const {
    MoveList Out; // at most one or two entries are written before the generator returns
    return genLegal<true>(Out);
}

// AnyOnly: return true as soon as one legal move exists (nothing useful is left in Out).
template <bool AnyOnly>
bool State::genLegal(MoveList& Out) const {
    Out.Size = 0;
    const Color Us = Side, Them = ~Side;
    const int K = KingSq[Us];
    const int KF = fileOf(K), KR = rankOf(K);
    GenCtx Ctx;
    std::memset(Ctx.PinDir, -1, sizeof(Ctx.PinDir));

    // ---- checkers and pins, scanning outward from the king
    int NumCheckers = 0;
    bool Between[NumSquares];
    std::memset(Between, 0, sizeof(Between));
    int CheckerSq = -1;
    for (int D = 0; D < 8; ++D) {
        int F = KF + kDF[D], R = KR + kDR[D];
        const uint8_t Back = (uint8_t)(1u << ((D + 4) & 7));
        int OwnSq = -1;
        bool First = true;
        int Path[8], NPath = 0;
        while (onBoard(F, R)) {
            const int S2 = makeSquare(F, R);
            const Piece P = Board[S2];
            if (P) {
                if (colorOf(P) == Us) {
                    if (OwnSq >= 0) break; // two own pieces: no pin on this line
                    OwnSq = S2;
                } else {
                    const bool Slides = slideMask(P) & Back;
                    if (OwnSq < 0) {
                        if (Slides || (First && (stepMask(P) & Back))) {
                            ++NumCheckers;
                            CheckerSq = S2;
                            if (Slides)
                                for (int I = 0; I < NPath; ++I) Between[Path[I]] = true;
                        }
                    } else if (Slides) {
                        Ctx.PinDir[OwnSq] = (int8_t)D;
                    }
                    break;
                }
            } else if (OwnSq < 0) {
                Path[NPath++] = S2;
            }
            First = false;
            F += kDF[D];
            R += kDR[D];
        }
    }
    {
        const int NR = KR - 2 * forward(Them); // an enemy knight here checks the king
        for (int DF = -1; DF <= 1; DF += 2)
            if (onBoard(KF + DF, NR) && Board[makeSquare(KF + DF, NR)] == makePiece(Them, Knight)) {
                ++NumCheckers;
                CheckerSq = makeSquare(KF + DF, NR);
            }
    }

    // ---- king moves
    for (int D = 0; D < 8; ++D) {
        const int F = KF + kDF[D], R = KR + kDR[D];
        if (!onBoard(F, R)) continue;
        const int To = makeSquare(F, R);
        const Piece P = Board[To];
        if (P && colorOf(P) == Us) continue;
        if (attackedWithout(To, Them, K, -1)) continue;
        Out.push(Move::make(K, To, false, King, typeOf(P)));
    }
    if (AnyOnly && Out.Size > 0) return true;
    if (NumCheckers >= 2) return Out.Size > 0;

    const bool InCheck = NumCheckers == 1;
    for (int S2 = 0; S2 < NumSquares; ++S2) Ctx.Target[S2] = !InCheck || Between[S2] || S2 == CheckerSq;

    // ---- board moves of the other pieces
    for (int From = 0; From < NumSquares; ++From) {
        const Piece P = Board[From];
        if (!P || colorOf(P) != Us || typeOf(P) == King) continue;
        const PieceType T = typeOf(P);
        const int Pin = Ctx.PinDir[From];
        const int FF = fileOf(From), FR = rankOf(From);
        if (T == Knight) {
            if (Pin >= 0) continue;
            const int R = FR + 2 * forward(Us);
            for (int DF = -1; DF <= 1; DF += 2) {
                if (!onBoard(FF + DF, R)) continue;
                const int To = makeSquare(FF + DF, R);
                const Piece Q = Board[To];
                if ((Q && colorOf(Q) == Us) || !Ctx.Target[To]) continue;
                pushMoves(Out, Us, From, To, T, typeOf(Q));
            }
            if (AnyOnly && Out.Size > 0) return true;
            continue;
        }
        const uint8_t Steps = stepMask(P), Slides = slideMask(P);
        for (int D = 0; D < 8; ++D) {
            const uint8_t Bit = (uint8_t)(1u << D);
            if (!((Steps | Slides) & Bit)) continue;
            if (AnyOnly && Out.Size > 0) return true;
        if (Pin >= 0 && (D & 3) != (Pin & 3)) continue; // only along the pin line
            int F = FF + kDF[D], R = FR + kDR[D];
            while (onBoard(F, R)) {
                const int To = makeSquare(F, R);
                const Piece Q = Board[To];
                if (Q && colorOf(Q) == Us) break;
                if (Ctx.Target[To]) pushMoves(Out, Us, From, To, T, typeOf(Q));
                if (Q || !(Slides & Bit)) break;
                F += kDF[D];
                R += kDR[D];
            }
        }
    }

    // ---- drops
    bool HaveHand = false;
    for (int T = Pawn; T <= Gold; ++T) HaveHand |= Hands[Us][T] != 0;
    if (AnyOnly && Out.Size > 0) return true;
    if (!HaveHand) return Out.Size > 0;
    if (InCheck) {
        bool Any = false;
        for (int S2 = 0; S2 < NumSquares; ++S2) Any |= Between[S2];
        if (!Any) return Out.Size > 0; // contact or knight check: a drop cannot help
    }
    bool PawnOnFile[9] = {false};
    if (Hands[Us][Pawn])
        for (int F = 0; F < 9; ++F)
            for (int R = 0; R < 9; ++R)
                if (Board[makeSquare(F, R)] == makePiece(Us, Pawn)) PawnOnFile[F] = true;
    const int EnemyKing = KingSq[Them];
    for (int To = 0; To < NumSquares; ++To) {
        if (Board[To]) continue;
        if (InCheck && !Between[To]) continue;
        const int Rel = relRank(Us, rankOf(To));
        for (int T = Pawn; T <= Gold; ++T) {
            if (!Hands[Us][T]) continue;
            if ((T == Pawn || T == Lance) && Rel == 0) continue;
            if (T == Knight && Rel <= 1) continue;
            if (T == Pawn) {
                if (PawnOnFile[fileOf(To)]) continue; // nifu
                if (EnemyKing >= 0 && To + forward(Us) == EnemyKing && fileOf(To) == fileOf(EnemyKing) &&
                    isPawnDropMate(To))
                    continue; // uchifuzume
            }
            Out.push(Move::makeDrop((PieceType)T, To));
            if (AnyOnly) return true;
        }
    }
    return Out.Size > 0;
}

// A pawn dropped on `To` checks the enemy king.  It is an illegal "drop pawn mate"
// iff the opponent then has no legal move.  (The reply generator cannot recurse: the
// checker is a contact pawn, so the replies contain no drops.)
bool State::isPawnDropMate(int To) const {
    State Tmp(0); // board-only copy: the (possibly long) move history is not needed
    std::memcpy(Tmp.Board, Board, sizeof(Board));
    std::memcpy(Tmp.Hands, Hands, sizeof(Hands));
    Tmp.KingSq[0] = KingSq[0];
    Tmp.KingSq[1] = KingSq[1];
    Tmp.put(To, makePiece(Side, Pawn));
    --Tmp.Hands[Side][Pawn];
    Tmp.Side = ~Side;
    return !Tmp.hasLegalMove();
}

void State::generatePseudo(MoveList& Out) const {
    Out.Size = 0;
    const Color Us = Side;
    for (int From = 0; From < NumSquares; ++From) {
        const Piece P = Board[From];
        if (!P || colorOf(P) != Us) continue;
        const PieceType T = typeOf(P);
        const int FF = fileOf(From), FR = rankOf(From);
        if (T == Knight) {
            const int R = FR + 2 * forward(Us);
            for (int DF = -1; DF <= 1; DF += 2) {
                if (!onBoard(FF + DF, R)) continue;
                const int To = makeSquare(FF + DF, R);
                const Piece Q = Board[To];
                if (Q && colorOf(Q) == Us) continue;
                pushMoves(Out, Us, From, To, T, typeOf(Q));
            }
            continue;
        }
        const uint8_t Steps = stepMask(P), Slides = slideMask(P);
        for (int D = 0; D < 8; ++D) {
            const uint8_t Bit = (uint8_t)(1u << D);
            if (!((Steps | Slides) & Bit)) continue;
            int F = FF + kDF[D], R = FR + kDR[D];
            while (onBoard(F, R)) {
                const int To = makeSquare(F, R);
                const Piece Q = Board[To];
                if (Q && colorOf(Q) == Us) break;
                if (T == King) Out.push(Move::make(From, To, false, King, typeOf(Q)));
                else pushMoves(Out, Us, From, To, T, typeOf(Q));
                if (Q || !(Slides & Bit)) break;
                F += kDF[D];
                R += kDR[D];
            }
        }
    }
    for (int To = 0; To < NumSquares; ++To) {
        if (Board[To]) continue;
        const int Rel = relRank(Us, rankOf(To));
        for (int T = Pawn; T <= Gold; ++T) {
            if (!Hands[Us][T]) continue;
            if ((T == Pawn || T == Lance) && Rel == 0) continue;
            if (T == Knight && Rel <= 1) continue;
            if (T == Pawn) {
                bool Nifu = false;
                for (int R = 0; R < 9; ++R) Nifu |= Board[makeSquare(fileOf(To), R)] == makePiece(Us, Pawn);
                if (Nifu) continue;
            }
            Out.push(Move::makeDrop((PieceType)T, To));
        }
    }
}

// Slow, independently structured generator: pseudo-legal moves filtered by actually
// playing them (own king must not be attacked; a checking pawn drop must leave the
// opponent at least one pseudo-legal reply that survives the same test).
void State::generateLegalMovesSlow(MoveList& Out) {
    MoveList Pseudo;
    generatePseudo(Pseudo);
    Out.Size = 0;
    const Color Us = Side;
    for (const Move& M : Pseudo) {
        doMove(M);
        bool Ok = !isAttacked(KingSq[Us], ~Us);
        if (Ok && M.isDrop() && M.moved() == Pawn && inCheck()) {
            MoveList Replies;
            generatePseudo(Replies);
            bool AnyReply = false;
            const Color Them = Side;
            for (const Move& R : Replies) {
                doMove(R);
                const bool Safe = !isAttacked(KingSq[Them], ~Them);
                undoMove();
                if (Safe) {
                    AnyReply = true;
                    break;
                }
            }
            Ok = AnyReply;
        }
        undoMove();
        if (Ok) Out.push(M);
    }
}

void State::doMove(Move M) {
    Undo U;
    U.M = M;
    U.HashBefore = hash();
    const Color Us = Side;
    const int To = M.to();
    if (M.isDrop()) {
        const PieceType T = M.moved();
        --Hands[Us][T];
        HandHash ^= zob().Hand[Us][T][Hands[Us][T]];
        put(To, makePiece(Us, T));
    } else {
        const int From = M.from();
        if (Board[To]) {
            const PieceType Cap = demote(typeOf(Board[To]));
            remove(To);
            HandHash ^= zob().Hand[Us][Cap][Hands[Us][Cap]];
            ++Hands[Us][Cap];
        }
        const PieceType T = M.promote() ? promote(M.moved()) : M.moved();
        remove(From);
        put(To, makePiece(Us, T));
    }
    Side = ~Side;
    U.WasCheck = KingSq[Side] >= 0 && isAttacked(KingSq[Side], Us);
    History.push_back(U);
}

void State::undoMove() {
    const Undo U = History.back();
    History.pop_back();
    Side = ~Side;
    const Color Us = Side;
    const Move M = U.M;
    const int To = M.to();
    if (M.isDrop()) {
        remove(To);
        HandHash ^= zob().Hand[Us][M.moved()][Hands[Us][M.moved()]];
        ++Hands[Us][M.moved()];
    } else {
        remove(To);
        put(M.from(), makePiece(Us, M.moved()));
        if (M.captured() != PT_None) {
            const PieceType Cap = demote(M.captured());
            --Hands[Us][Cap];
            HandHash ^= zob().Hand[Us][Cap][Hands[Us][Cap]];
            put(To, makePiece(~Us, M.captured()));
        }
    }
}

Move State::moveFrom16(uint16_t M16) const {
    const int To = M16 & 127, From = (M16 >> 7) & 127;
    const bool Promo = (M16 >> 14) & 1;
    if (From >= 81) return Move::makeDrop((PieceType)(From - 81), To);
    return Move::make(From, To, Promo, typeOf(Board[From]), typeOf(Board[To]));
}

RepetitionStatus State::repetitionStatus(bool CheckPerpetual) const {
    const uint64_t H = hash();
    const int N2 = (int)History.size();
    int Count = 0;
    int Earliest = -1;
    // History[i].HashBefore is the position before move i; same side to move every 2 plies
    for (int I = N2 - 2; I >= 0; I -= 2) {
        if (History[I].HashBefore == H) {
            ++Count;
            Earliest = I;
            if (Count >= 3) break;
        }
    }
    if (Count < 3) return NoRepetition;
    if (CheckPerpetual) {
        // moves Earliest .. N2-1; the side to move now made the moves at Earliest, Earliest+2, ...
        bool MineAllChecks = true, TheirsAllChecks = true;
        for (int I = Earliest; I < N2; ++I) {
            if (((I - Earliest) & 1) == 0) MineAllChecks &= History[I].WasCheck;
            else TheirsAllChecks &= History[I].WasCheck;
        }
        if (TheirsAllChecks) return WinRepetition;  // the opponent checked perpetually: they lose
        if (MineAllChecks) return LossRepetition;
    }
    return Repetition;
}

bool State::canDeclare() const {
    const Color Us = Side;
    const int K = KingSq[Us];
    if (K < 0 || !inZone(Us, rankOf(K))) return false;
    if (inCheck()) return false;
    int Pieces = 0, Points = 0;
    for (int S2 = 0; S2 < NumSquares; ++S2) {
        const Piece P = Board[S2];
        if (!P || colorOf(P) != Us || typeOf(P) == King || !inZone(Us, rankOf(S2))) continue;
        ++Pieces;
        const PieceType B = demote(typeOf(P));
        Points += (B == Bishop || B == Rook) ? 5 : 1;
    }
    if (Pieces < 10) return false;
    for (int T = Pawn; T <= Gold; ++T) Points += Hands[Us][T] * ((T == Bishop || T == Rook) ? 5 : 1);
    return Points >= (Us == Black ? 28 : 27);
}

namespace {
inline int sgn(int V) { return (V > 0) - (V < 0); }
inline int dirIndex(int SF, int SR) {
    for (int D = 0; D < 8; ++D)
        if (kDF[D] == SF && kDR[D] == SR) return D;
    return -1;
}
inline bool onLine(int DF, int DR) { return (DF || DR) && (DF == 0 || DR == 0 || DF == DR || DF == -DR); }
} // namespace

bool State::givesCheck(Move M) const {
    const Color Us = Side;
    const int K = KingSq[~Us], KF = fileOf(K), KR = rankOf(K);
    const int To = M.to(), TF = fileOf(To), TR = rankOf(To);
    const PieceType T = M.promote() ? promote(M.moved()) : M.moved();
    const Piece P = makePiece(Us, T);
    const int From = M.isDrop() ? -1 : M.from();
    // ---- direct: the piece as it stands on To attacks the king
    if (T == Knight) {
        if (KR - TR == 2 * forward(Us) && (KF - TF == 1 || KF - TF == -1)) return true;
    } else {
        const int DF = KF - TF, DR = KR - TR;
        if (onLine(DF, DR)) {
            const int SF = sgn(DF), SR = sgn(DR);
            const uint8_t Bit = (uint8_t)(1u << dirIndex(SF, SR));
            const int Dist = DF ? (DF < 0 ? -DF : DF) : (DR < 0 ? -DR : DR);
            if (Dist == 1 && (stepMask(P) & Bit)) return true;
            if (slideMask(P) & Bit) {
                bool Clear = true;
                for (int F = TF + SF, R = TR + SR; F != KF || R != KR; F += SF, R += SR) {
                    const int S2 = makeSquare(F, R);
                    if (S2 != From && Board[S2]) {
                        Clear = false;
                        break;
                    }
                }
                if (Clear) return true;
            }
        }
    }
    // ---- discovered: From leaves a line between the king and one of our sliders
    if (From >= 0) {
        const int DF = fileOf(From) - KF, DR = rankOf(From) - KR;
        if (onLine(DF, DR)) {
            const int SF = sgn(DF), SR = sgn(DR);
            const int TDF = TF - KF, TDR = TR - KR;
            const bool StaysOnRay = onLine(TDF, TDR) && sgn(TDF) == SF && sgn(TDR) == SR;
            if (!StaysOnRay) {
                int F = KF + SF, R = KR + SR;
                bool Clear = true;
                for (; makeSquare(F, R) != From; F += SF, R += SR)
                    if (Board[makeSquare(F, R)]) {
                        Clear = false;
                        break;
                    }
                if (Clear) {
                    const uint8_t Back = (uint8_t)(1u << ((dirIndex(SF, SR) + 4) & 7)); // from the slider toward the king
                    for (F += SF, R += SR; onBoard(F, R); F += SF, R += SR) {
                        const Piece Q = Board[makeSquare(F, R)];
                        if (!Q) continue;
                        if (colorOf(Q) == Us && (slideMask(Q) & Back)) return true;
                        break;
                    }
                }
            }
        }
    }
    return false;
}

void State::checkInfo(CheckInfo& CI) const {
    const int K = KingSq[~Side], KF = fileOf(K), KR = rankOf(K);
    std::memset(CI.RayDir, -1, sizeof(CI.RayDir));
    std::memset(CI.Adjacent, 0, sizeof(CI.Adjacent));
    std::memset(CI.OnLine, 0, sizeof(CI.OnLine));
    for (int D = 0; D < 8; ++D) {
        bool Seen = true, First = true;
        for (int F = KF + kDF[D], R = KR + kDR[D]; onBoard(F, R); F += kDF[D], R += kDR[D]) {
            const int S2 = makeSquare(F, R);
            CI.OnLine[S2] = true;
            if (Seen) {
                CI.RayDir[S2] = (int8_t)D;
                CI.Adjacent[S2] = First;
                if (Board[S2]) Seen = false; // the first occupied square is still a landing square (a capture)
            }
            First = false;
        }
    }
    // a knight of the side to move on (KF -+ 1, KR - 2*forward(Us)) attacks the king
    const int NR = KR - 2 * forward(Side);
    CI.KnightSq[0] = onBoard(KF - 1, NR) ? makeSquare(KF - 1, NR) : -1;
    CI.KnightSq[1] = onBoard(KF + 1, NR) ? makeSquare(KF + 1, NR) : -1;
}

bool State::givesCheck(Move M, const CheckInfo& CI) const {
    // a piece that leaves one of the king's lines may discover a check, or slide along the line: exact test
    if (!M.isDrop() && CI.OnLine[M.from()]) return givesCheck(M);
    const int To = M.to();
    const PieceType T = M.promote() ? promote(M.moved()) : M.moved();
    if (T == Knight) return To == CI.KnightSq[0] || To == CI.KnightSq[1];
    const int D = CI.RayDir[To];
    if (D < 0) return false;
    // (the squares between the king and To are empty now and stay empty: From is not on this ray)
    const Piece P = makePiece(Side, T);
    const uint8_t Back = (uint8_t)(1u << ((D + 4) & 7)); // direction from To toward the king
    return (slideMask(P) & Back) || (CI.Adjacent[To] && (stepMask(P) & Back));
}

// A square the king of `Defender` can step to safely (empty or enemy-occupied, not attacked once the king
// has left its square), or -1.  The cheap way to see that a check is not mate: most checks leave one.
int State::kingFlight(Color Defender) const {
    const int K = KingSq[Defender];
    const int F0 = fileOf(K), R0 = rankOf(K);
    for (int D = 0; D < 8; ++D) {
        const int F = F0 + kDF[D], R = R0 + kDR[D];
        if (!onBoard(F, R)) continue;
        const int Sq = makeSquare(F, R);
        const Piece P = Board[Sq];
        if (P && colorOf(P) == Defender) continue;
        if (!attackedWithout(Sq, ~Defender, K, -1)) return Sq;
    }
    return -1;
}

// Mate in one for the side to move, without the full legal move list: pseudo-legal moves, kept only if
// they give check (exact, no move made), and only those are played -- the own king must then be safe
// (legality), the opponent's king without a flight square, and the opponent without any reply.  A mate
// by a dropped pawn is not a legal move.
Move State::findMateInOneQuick() {
    MoveList Pseudo;
    generatePseudo(Pseudo);
    const Color Us = Side;
    CheckInfo CI;
    checkInfo(CI);
    for (const Move& M : Pseudo) {
        if (!givesCheck(M, CI)) continue;
        if (M.isDrop() && M.moved() == Pawn) continue;
        doMove(M);
        bool Mate = false;
        if (!isAttacked(KingSq[Us], ~Us) && kingFlight(~Us) < 0) { // a legal move, and the king cannot just step away
            Mate = !hasLegalMove();
        }
        undoMove();
        if (Mate) return M;
    }
    return Move();
}

Move State::findMate(int Depth, bool Prefilter, const MoveList* Legal) {
    if (Depth < 1) return Move();
    MoveList Own;
    if (!Legal) generateLegalMoves(Own);
    const MoveList& Moves = Legal ? *Legal : Own;
    // First pass over the checking moves: mate in one.  A check that leaves the king a flight square is
    // not mate, and its replies are not generated yet (Flight[] remembers the square).  Second pass
    // (Depth >= 3): a check mates in three if every reply runs into a mate in one -- the king's step to the
    // remembered flight square is tried first, and only a check that survives it gets its full reply list.
    Move Checks[600];
    int8_t Flight[600];
    int NumChecks = 0;
    CheckInfo CI;
    if (Prefilter) checkInfo(CI);
    for (const Move& M : Moves) {
        if (Prefilter && !givesCheck(M, CI)) continue;
        doMove(M);
        if (!inCheck()) { // (after the move the side to move is the defender; only reached without the prefilter)
            undoMove();
            continue;
        }
        const int Fl = Prefilter ? kingFlight(Side) : -1;
        int NumReplies = 1;
        if (Fl < 0) NumReplies = hasLegalMove() ? 1 : 0;
        undoMove();
        if (NumReplies == 0) return M; // drop-pawn mate is not a legal move, so M is a real mate
        Flight[NumChecks] = (int8_t)Fl;
        Checks[NumChecks++] = M;
    }
    if (Depth < 3) return Move();
    for (int I = 0; I < NumChecks; ++I) {
        doMove(Checks[I]);
        bool AllMated = true;
        if (Flight[I] >= 0) { // the cheapest refutation candidate: the king steps away
            const int K = KingSq[Side];
            doMove(Move::make(K, Flight[I], false, King, typeOf(Board[Flight[I]])));
            AllMated = !findMateInOneQuick().isNone();
            undoMove();
        }
        if (AllMated) {
            MoveList Replies;
            generateLegalMoves(Replies);
            for (const Move& R : Replies) {
                doMove(R);
                const bool Mated = Prefilter ? !findMateInOneQuick().isNone() : !findMate(1, false).isNone();
                undoMove();
                if (!Mated) {
                    AllMated = false;
                    break;
                }
            }
        }
        undoMove();
        if (AllMated) return Checks[I];
    }
    return Move();
}

uint64_t State::perft(int Depth) {
    MoveList L;
    generateLegalMoves(L);
    if (Depth <= 1) return (uint64_t)L.size();
    uint64_t Nodes = 0;
    for (const Move& M : L) {
        doMove(M);
        Nodes += perft(Depth - 1);
        undoMove();
    }
    return Nodes;
}

std::string moveToUsi(Move M) {
    auto Sq = [](int S2) {
        std::string R;
        R += (char)('1' + fileOf(S2));
        R += (char)('a' + rankOf(S2));
        return R;
    };
    if (M.isDrop()) return std::string(1, kSfenPiece[M.moved()]) + "*" + Sq(M.to());
    return Sq(M.from()) + Sq(M.to()) + (M.promote() ? "+" : "");
}

} // namespace shogi
} // namespace engine
} // namespace nshogi
