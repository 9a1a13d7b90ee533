// shogi.h -- a compact shogi rules core for the self-play path.
//
// The reference delegates all rules to the un-vendored libnshogi
// (core::State, core::MoveGenerator, RepetitionStatus, canDeclare ...; call sites:
// /root/reference/src/selfplay/worker.cc:112-381,477-526, src/mcts/searchworker.cc).
// libnshogi is not available here, so this is an independent implementation of the
// rules of shogi, pinned by the game's public perft numbers (tests/test_shogi_core.py)
// and by a slow make/unmake cross-check generator; it is NOT a restatement of
// libnshogi and its internal conventions (square numbering, move encoding) are
// this build's own.
//
// Conventions: Black (sente) moves toward rank 0.  Square = file*9 + rank,
// file 0 = "1" (Black's right), rank 0 = "a" (Black's far side).
#ifndef NSG_SHOGI_H
#define NSG_SHOGI_H

#include <cstdint>
#include <string>
#include <vector>

namespace nshogi {
namespace engine {
namespace shogi {

enum Color : uint8_t { Black = 0, White = 1, NoColor = 2 };
inline Color operator~(Color C) { return (Color)(C ^ 1); }

enum PieceType : uint8_t {
    PT_None = 0, Pawn = 1, Lance = 2, Knight = 3, Silver = 4, Bishop = 5, Rook = 6, Gold = 7, King = 8,
    ProPawn = 9, ProLance = 10, ProKnight = 11, ProSilver = 12, Horse = 13, Dragon = 14, PT_Num = 15
};
constexpr int NumSquares = 81;
constexpr int NumHandTypes = 7; // Pawn .. Gold (types 1..7)

inline bool canPromoteType(PieceType T) { return T >= Pawn && T <= Rook; }
inline bool isPromoted(PieceType T) { return T >= ProPawn; }
inline PieceType promote(PieceType T) { return (PieceType)(T + 8); }
inline PieceType demote(PieceType T) { return isPromoted(T) ? (PieceType)(T - 8) : T; }

// piece on a square: 0 = empty, else (color << 4) | type
using Piece = uint8_t;
inline Piece makePiece(Color C, PieceType T) { return (Piece)((C << 4) | T); }
inline Color colorOf(Piece P) { return (Color)(P >> 4); }
inline PieceType typeOf(Piece P) { return (PieceType)(P & 15); }

inline int fileOf(int Sq) { return Sq / 9; }
inline int rankOf(int Sq) { return Sq % 9; }
inline int makeSquare(int File, int Rank) { return File * 9 + Rank; }

// 32-bit move: to[0:7) from[7:14) (81 + type for drops) promote[14] moved type[15:19)
// captured type[19:23).  The low 16 bits identify the move in a position.
struct Move {
    uint32_t V = 0;
    static Move make(int From, int To, bool Promote, PieceType Moved, PieceType Captured) {
        Move M;
        M.V = (uint32_t)To | ((uint32_t)From << 7) | ((uint32_t)Promote << 14) |
              ((uint32_t)Moved << 15) | ((uint32_t)Captured << 19);
        return M;
    }
    static Move makeDrop(PieceType T, int To) { return make(81 + T, To, false, T, PT_None); }
    int to() const { return (int)(V & 127); }
    int from() const { return (int)((V >> 7) & 127); }
    bool isDrop() const { return from() >= 81; }
    bool promote() const { return (V >> 14) & 1; }
    PieceType moved() const { return (PieceType)((V >> 15) & 15); } // type BEFORE promotion
    PieceType captured() const { return (PieceType)((V >> 19) & 15); }
    uint16_t move16() const { return (uint16_t)(V & 0x7fff); }
    bool isNone() const { return V == 0; }
    bool operator==(const Move& O) const { return V == O.V; }
};

struct MoveList {
    Move Moves[600]; // the maximum number of legal moves in any shogi position is 593
    int Size = 0;
    void push(Move M) { Moves[Size++] = M; }
    int size() const { return Size; }
    const Move& operator[](int I) const { return Moves[I]; }
    const Move* begin() const { return Moves; }
    const Move* end() const { return Moves + Size; }
};

enum RepetitionStatus : uint8_t { NoRepetition = 0, Repetition = 1, WinRepetition = 2, LossRepetition = 3 };

// Role of core::StateConfig (selfplay/worker.cc:132-150).
struct StateConfig {
    uint16_t MaxPly = 1024;
    float BlackDrawValue = 0.5f;
    float WhiteDrawValue = 0.5f;
    bool Declare27 = true;
};

class State {
 public:
    State(); // the initial position
    static State fromSfen(const std::string& Sfen);
    std::string toSfen() const;

    Color sideToMove() const { return Side; }
    int ply() const { return (int)History.size() + PlyOffset; }
    Piece pieceOn(int Sq) const { return Board[Sq]; }
    int hand(Color C, PieceType T) const { return Hands[C][T]; }
    int kingSquare(Color C) const { return KingSq[C]; }
    uint64_t hash() const { return BoardHash ^ HandHash ^ (Side == White ? SideKey : 0); }

    void doMove(Move M);
    void undoMove();
    Move lastMove() const { return History.empty() ? Move() : History.back().M; }

    // Legal moves (pins, checks, nifu, dead-square drops and drop-pawn-mate excluded).
    void generateLegalMoves(MoveList& Out) const;
    // Does the side to move have any legal move?  The same generator, stopping at the first move
    // (the mate search only asks whether a checked side has a reply at all).
    bool hasLegalMove() const;
    // Independent slow generator: pseudo-legal + make/unmake king test (cross-check).
    void generateLegalMovesSlow(MoveList& Out);

    bool inCheck() const { return isAttacked(KingSq[Side], ~Side); }
    bool isAttacked(int Sq, Color By) const { return attackedWithout(Sq, By, -1, -1); }

    // Fourfold repetition.  With CheckPerpetual, a repetition whose every move by one
    // side was a check is a loss for that side: Win/Loss are from the side to move.
    RepetitionStatus repetitionStatus(bool CheckPerpetual) const;
    // 27-point declaration win (nyugyoku sengen) for the side to move.
    bool canDeclare() const;

    Move moveFrom16(uint16_t M16) const; // rebuild moved/captured from the position

    // Forced-mate search by checks only (role of libnshogi's solver::dfs::solve(State, 3) at
    // selfplay/worker.cc:349-358): a move of the side to move that checkmates within Depth
    // plies (1 or 3) against every defence, or a none move.  Prefilter = false tests every
    // legal move for check (slow reference for the tests).
    Move findMate(int Depth, bool Prefilter = true, const MoveList* Legal = nullptr); // Legal: the position's legal moves, if already generated
    // Mate in one from pseudo-legal checking moves only (the inner test of findMate(3)).
    Move findMateInOneQuick();
    int kingFlight(Color Defender) const;
    // Does this legal move of the side to move give check (directly or by discovery)?  Exact,
    // from the current position, without making the move.
    bool givesCheck(Move M) const;
    // The squares from which the enemy king can be checked, worked out once per position for the
    // many givesCheck questions of a mate search: RayDir[sq] = direction (0..7) of the ray from the
    // king on which sq is visible (through empty squares; the first occupied square included), or -1;
    // Adjacent[sq]: one step from the king; OnLine[sq]: on one of the king's eight lines at all, seen or
    // not (a piece leaving such a square may discover a check or slide along the line: those moves take
    // the exact test); KnightSq: where a knight of the side to move would check.
    struct CheckInfo {
        int8_t RayDir[NumSquares];
        bool Adjacent[NumSquares];
        bool OnLine[NumSquares];
        int KnightSq[2];
    };
    void checkInfo(CheckInfo& CI) const;
    bool givesCheck(Move M, const CheckInfo& CI) const; // == givesCheck(M)

    uint64_t perft(int Depth);

 private:
    explicit State(int); // empty board (fromSfen)
    struct Undo {
        Move M;
        uint64_t HashBefore;  // position hash before the move
        bool WasCheck;        // the move gave check
    };
    bool attackedWithout(int Sq, Color By, int RemovedSq, int AddedSq) const;
    bool isPawnDropMate(int To) const;
    void put(int Sq, Piece P);
    void remove(int Sq);
    void generatePseudo(MoveList& Out) const;
    template <bool AnyOnly> bool genLegal(MoveList& Out) const;

    Piece Board[NumSquares];
    uint8_t Hands[2][8];
    int KingSq[2];
    Color Side;
    int PlyOffset;
    uint64_t BoardHash, HandHash;
    std::vector<Undo> History;
    static uint64_t SideKey;
};

std::string moveToUsi(Move M);

} // namespace shogi
} // namespace engine
} // namespace nshogi

#endif
