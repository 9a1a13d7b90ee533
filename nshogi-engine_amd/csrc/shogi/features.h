// features.h -- feature planes and policy move index for the self-play path.
//
// Plays the role of libnshogi's ml::FeatureStackComptime<...86 x FT_*...>
// (the preset /root/reference/src/evaluate/preset.h:20-66, used at
// src/selfplay/evaluationworker.cc:87-92 via FeatureType::constructAt) and of
// ml::getMoveIndex<ChannelsFirst> (src/selfplay/frame.cc:102-105).  libnshogi is
// absent, so the plane SEMANTICS below are this build's reading of the feature
// names -- parity with libnshogi is unpinned; the 16-byte
// FeatureBitboard LAYOUT is the reference's (src/cuda/extractbit.cu:20-37).
//
// Plane order = the order of preset.h.  Everything is relative to the side to
// move ("My" / "Op"); when White moves the rotate flag is set so the expansion
// kernel turns the board by 180 degrees.
//   0..13   My  Pawn Lance Knight Silver Gold King Bishop Rook +P +L +N +S Horse Dragon  (value 1 on occupied squares)
//   14..27  Op  (same order)
//   28..53  My hand: Pawn>=1..6, Lance>=1..4, Knight>=1..4, Silver>=1..4, Gold>=1..4, Bishop>=1..2, Rook>=1..2 (all squares)
//   54..79  Op hand (same order)
//   80, 81  Black to move, White to move (all squares)
//   82      Progress      = ply / MaxPly          (scalar on all squares)
//   83      ProgressUnit  = 1 / MaxPly
//   84, 85  MyDrawValue, OpDrawValue
#ifndef NSG_SHOGI_FEATURES_H
#define NSG_SHOGI_FEATURES_H

#include "shogi.h"

#include <cstdint>

namespace nshogi {
namespace engine {
namespace shogi {

constexpr int NumFeaturePlanes = 86;
constexpr int MoveIndexMax = 27 * 81;

struct FeaturePlane { // == ml::FeatureBitboard (16 bytes)
    uint64_t Lo, Hi;
};

// Writes NumFeaturePlanes planes (FeatureType::constructAt).
void buildFeatures(const State& S, const StateConfig& Config, FeaturePlane* Out);

// Policy index of a move for the side to move: class*81 + destination square, both in
// the mover's own orientation.  Classes: 8 directions + 2 knight jumps (0..9), the same
// with promotion (10..19), drops of Pawn..Gold (20..26).
int moveIndex(Color SideToMove, Move M);

} // namespace shogi
} // namespace engine
} // namespace nshogi

#endif
