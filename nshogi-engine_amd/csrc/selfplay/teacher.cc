// teacher.cc -- see teacher.h
#include "teacher.h"

#include <cstring>
#include <stdexcept>

namespace nshogi {
namespace engine {
namespace selfplay {

TeacherWriter::TeacherWriter(const std::string& Path) {
    Out = std::fopen(Path.c_str(), "wb");
    if (Out == nullptr) throw std::runtime_error("cannot open teacher file " + Path);
    const uint32_t Header[4] = {0x5447534eu /* "NSGT" */, 1u, (uint32_t)sizeof(TeacherRecord), 0u};
    std::fwrite(Header, sizeof(Header), 1, Out);
}

TeacherWriter::~TeacherWriter() {
    if (Out) std::fclose(Out);
}

std::size_t TeacherWriter::saveGame(const std::vector<uint32_t>& Moves, const std::vector<uint8_t>& FullSearch,
                                    const shogi::StateConfig& Config, shogi::Color Winner) {
    std::vector<TeacherRecord> Buf;
    shogi::State Replay;
    for (std::size_t Ply = 0; Ply < Moves.size(); ++Ply) {
        shogi::Move M;
        M.V = Moves[Ply];
        if (Ply < FullSearch.size() && FullSearch[Ply]) { // saveworker.cc:172-178
            TeacherRecord R;
            std::memset(&R, 0, sizeof(R));
            for (int Sq = 0; Sq < shogi::NumSquares; ++Sq) R.Board[Sq] = Replay.pieceOn(Sq);
            for (int C = 0; C < 2; ++C)
                for (int T = 0; T < shogi::NumHandTypes; ++T)
                    R.Hands[C][T] = (uint8_t)Replay.hand((shogi::Color)C, (shogi::PieceType)(T + 1));
            R.SideToMove = (uint8_t)Replay.sideToMove();
            R.Winner = (uint8_t)Winner;
            R.Declare27 = Config.Declare27 ? 1 : 0;
            R.Ply = (uint16_t)Replay.ply();
            R.NextMove16 = M.move16();
            R.MaxPly = Config.MaxPly;
            R.BlackDrawValue = Config.BlackDrawValue;
            R.WhiteDrawValue = Config.WhiteDrawValue;
            R.GameLength = (uint16_t)Moves.size();
            Buf.push_back(R);
        }
        Replay.doMove(M);
    }
    if (!Buf.empty()) {
        std::lock_guard<std::mutex> Lock(Mutex);
        if (std::fwrite(Buf.data(), sizeof(TeacherRecord), Buf.size(), Out) != Buf.size())
            throw std::runtime_error("short write to the teacher file");
        Records += Buf.size();
    }
    return Buf.size();
}

GameLog::GameLog(const std::string& Path) {
    Out = std::fopen(Path.c_str(), "w");
    if (!Out) throw std::runtime_error("could not open the game log: " + Path);
}

GameLog::~GameLog() {
    if (Out) std::fclose(Out);
}

void GameLog::add(uint64_t GameId, shogi::Color Winner, const std::vector<uint32_t>& Moves) {
    uint64_t Digest = 0xcbf29ce484222325ULL; // FNV-1a over the move words
    std::string Line;
    for (uint32_t V : Moves) {
        Digest = (Digest ^ V) * 0x100000001b3ULL;
        shogi::Move M;
        M.V = V;
        Line += ' ';
        Line += shogi::moveToUsi(M);
    }
    std::lock_guard<std::mutex> Lock(Mutex);
    std::fprintf(Out, "%llu %d %zu %llu%s\n", (unsigned long long)GameId, (int)Winner, Moves.size(),
                 (unsigned long long)Digest, Line.c_str());
    std::fflush(Out);
}

LeafLog::LeafLog(const std::string& Path) {
    Out = std::fopen(Path.c_str(), "w");
    if (!Out) throw std::runtime_error("could not open the leaf log: " + Path);
}

LeafLog::~LeafLog() {
    if (Out) std::fclose(Out);
}

void LeafLog::add(uint64_t Group, uint64_t Batch, std::size_t Slot, const std::string& Sfen, uint16_t MaxPly, float BlackDraw,
                  const void* Planes, std::size_t Bytes) {
    static const char* Digits = "0123456789abcdef";
    std::string Hex;
    Hex.reserve(Bytes * 2);
    const unsigned char* P = static_cast<const unsigned char*>(Planes);
    for (std::size_t I = 0; I < Bytes; ++I) {
        Hex += Digits[P[I] >> 4];
        Hex += Digits[P[I] & 15];
    }
    std::lock_guard<std::mutex> Lock(Mutex);
    std::fprintf(Out, "%llu\t%llu\t%zu\t%s\t%u\t%.9g\t%s\n", (unsigned long long)Group, (unsigned long long)Batch, Slot,
                 Sfen.c_str(), (unsigned)MaxPly, (double)BlackDraw, Hex.c_str());
}

} // namespace selfplay
} // namespace engine
} // namespace nshogi
