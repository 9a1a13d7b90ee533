#include "features.h"

#include <cstring>

namespace nshogi {
namespace engine {
namespace shogi {

namespace {
constexpr uint64_t kAllLo = (1ULL << 63) - 1;
constexpr uint64_t kAllHi = (1ULL << 18) - 1;
constexpr uint64_t kOne = 0x3f800000ULL << 32;

inline uint64_t f32Bits(float V) {
    uint32_t U;
    std::memcpy(&U, &V, 4);
    return (uint64_t)U << 32;
}

// piece type -> plane offset inside the 14 board planes
constexpr int kBoardPlane[PT_Num] = {-1, 0, 1, 2, 3, 6, 7, 4, 5, 8, 9, 10, 11, 12, 13};
constexpr PieceType kHandOrder[7] = {Pawn, Lance, Knight, Silver, Gold, Bishop, Rook};
constexpr int kHandPlanes[7] = {6, 4, 4, 4, 4, 2, 2};
} // namespace

void buildFeatures(const State& S, const StateConfig& Config, FeaturePlane* Out) {
    const Color Us = S.sideToMove();
    const uint64_t Rot = (Us == White) ? (1ULL << 24) : 0;
    for (int P = 0; P < NumFeaturePlanes; ++P) {
        Out[P].Lo = 0;
        Out[P].Hi = Rot | kOne;
    }
    for (int Sq = 0; Sq < NumSquares; ++Sq) {
        const Piece Pc = S.pieceOn(Sq);
        if (!Pc) continue;
        const int Plane = (colorOf(Pc) == Us ? 0 : 14) + kBoardPlane[typeOf(Pc)];
        if (Sq < 63) Out[Plane].Lo |= 1ULL << Sq;
        else Out[Plane].Hi |= 1ULL << (Sq - 63);
    }
    int Plane = 28;
    for (int Side = 0; Side < 2; ++Side) {
        const Color C = Side == 0 ? Us : ~Us;
        for (int I = 0; I < 7; ++I) {
            const int Count = S.hand(C, kHandOrder[I]);
            for (int K = 1; K <= kHandPlanes[I]; ++K, ++Plane) {
                if (Count >= K) {
                    Out[Plane].Lo = kAllLo;
                    Out[Plane].Hi |= kAllHi;
                }
            }
        }
    }
    auto Fill = [&](int P, float Value) {
        Out[P].Lo = kAllLo;
        Out[P].Hi = Rot | kAllHi | f32Bits(Value);
    };
    if (Us == Black) Fill(80, 1.0f); else Fill(81, 1.0f);
    const float MaxPly = Config.MaxPly > 0 ? (float)Config.MaxPly : 1.0f;
    Fill(82, (float)S.ply() / MaxPly);
    Fill(83, 1.0f / MaxPly);
    Fill(84, Us == Black ? Config.BlackDrawValue : Config.WhiteDrawValue);
    Fill(85, Us == Black ? Config.WhiteDrawValue : Config.BlackDrawValue);
}

int moveIndex(Color SideToMove, Move M) {
    const int To = SideToMove == White ? 80 - M.to() : M.to();
    if (M.isDrop()) return (20 + (M.from() - 81) - 1) * 81 + To; // works for 16-bit moves too
    const int From = SideToMove == White ? 80 - M.from() : M.from();
    const int DF = fileOf(To) - fileOf(From), DR = rankOf(To) - rankOf(From);
    int Cls;
    if (DR == -2 && (DF == 1 || DF == -1)) {
        Cls = DF < 0 ? 8 : 9; // knight jumps
    } else if (DF == 0) {
        Cls = DR < 0 ? 0 : 5;
    } else if (DR == 0) {
        Cls = DF < 0 ? 3 : 4;
    } else if (DR < 0) {
        Cls = DF < 0 ? 1 : 2;
    } else {
        Cls = DF < 0 ? 6 : 7;
    }
    if (M.promote()) Cls += 10;
    return Cls * 81 + To;
}

} // namespace shogi
} // namespace engine
} // namespace nshogi
