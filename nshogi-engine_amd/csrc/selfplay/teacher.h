// teacher.h -- training-record writer of the self-play engine.
//
// Role of selfplay::SaveWorker::save (/root/reference/src/selfplay/saveworker.cc:160-182): when a
// game ends it is replayed from its initial position and, for every ply whose root search was a
// *full* search (playout-cap randomisation, worker.cc:171-199), one record is written holding the
// position, the game's StateConfig, the move played and the winner.  The reference serialises
// libnshogi's ml::SimpleTeacher through io::file::simple_teacher::save; that byte format lives in
// the absent library, so the file written here is this build's own ("NSGT" v1, below) carrying the
// same fields -- parity with libnshogi's teacher format is unpinned.
//
// File: 16-byte header {"NSGT", u32 version = 1, u32 record_size = 128, u32 reserved}, then
// fixed 128-byte little-endian records (struct TeacherRecord).  Squares, piece codes and move16
// follow csrc/shogi/shogi.h.  Reader: nshogi-engine_amd/teacher.py.
#ifndef NSG_SELFPLAY_TEACHER_H
#define NSG_SELFPLAY_TEACHER_H

#include "../shogi/shogi.h"

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

namespace nshogi {
namespace engine {
namespace selfplay {

#pragma pack(push, 1)
struct TeacherRecord {
    uint8_t Board[81];     // shogi::Piece per square ((color << 4) | type, 0 = empty), square = file*9 + rank
    uint8_t Hands[2][7];   // [color][Pawn..Gold] counts
    uint8_t SideToMove;    // 0 black, 1 white
    uint8_t Winner;        // 0 black, 1 white, 2 draw (of the whole game)
    uint8_t Declare27;     // StateConfig: declaration rule enabled
    uint16_t Ply;          // ply of this position (0 = initial position)
    uint16_t NextMove16;   // the move played here (shogi::Move::move16)
    uint16_t MaxPly;       // StateConfig
    float BlackDrawValue;  // StateConfig
    float WhiteDrawValue;
    uint16_t GameLength;   // plies in the finished game
    uint8_t Reserved[14];
};
#pragma pack(pop)
static_assert(sizeof(TeacherRecord) == 128, "teacher record is 128 bytes");

// One output file shared by the engines of a process; a finished game's records are appended
// under one lock (the reference funnels frames to a single SaveWorker thread, saveworker.cc:45-77).
class TeacherWriter {
 public:
    explicit TeacherWriter(const std::string& Path);
    ~TeacherWriter();
    TeacherWriter(const TeacherWriter&) = delete;
    TeacherWriter& operator=(const TeacherWriter&) = delete;

    // Replays `Moves` (full 32-bit shogi::Move values) from the initial position and writes a
    // record for every ply with FullSearch[ply] != 0.  Returns the number of records written.
    std::size_t saveGame(const std::vector<uint32_t>& Moves, const std::vector<uint8_t>& FullSearch,
                         const shogi::StateConfig& Config, shogi::Color Winner);
    uint64_t records() const { return Records; }

 private:
    std::FILE* Out = nullptr;
    std::mutex Mutex;
    uint64_t Records = 0;
};

// One text line per finished game: "<game id> <winner 0|1|2> <plies> <move digest> <usi moves...>".
// The game id is global to the run (slot + k * slots), so two runs that spread the same slots
// differently over threads, groups and GPUs can be compared game by game.
class GameLog {
 public:
    explicit GameLog(const std::string& Path);
    ~GameLog();
    GameLog(const GameLog&) = delete;
    GameLog& operator=(const GameLog&) = delete;
    void add(uint64_t GameId, shogi::Color Winner, const std::vector<uint32_t>& Moves);

 private:
    std::FILE* Out = nullptr;
    std::mutex Mutex;
};

// Test hook of the host batch packing (selfplay::EvaluationWorker::doTask's role): one text
// line per leaf of every batch -- engine group, batch number, slot, the leaf's SFEN, its StateConfig (MaxPly,
// BlackDrawValue) and the 1376 bytes found in that slot of the pinned batch buffer AFTER packing (hex) -- so a
// test can rebuild what the slot should hold from the SFEN alone.
class LeafLog {
 public:
    explicit LeafLog(const std::string& Path);
    ~LeafLog();
    LeafLog(const LeafLog&) = delete;
    LeafLog& operator=(const LeafLog&) = delete;
    void add(uint64_t Group, uint64_t Batch, std::size_t Slot, const std::string& Sfen, uint16_t MaxPly, float BlackDraw,
             const void* Planes, std::size_t Bytes);

 private:
    std::FILE* Out = nullptr;
    std::mutex Mutex;
};

} // namespace selfplay
} // namespace engine
} // namespace nshogi

#endif
