// selfplay_main.cc -- self-play driver: T engine threads per GPU, each with two
// executors (one per game group).  Prints one JSON line: games/sec per the
// reference's definition (finished games / elapsed, saveworker.cc:135-137), average
// batch size and cache-hit ratio (selfplayinfo.cc:51-57,72-78), plus playouts/s.
//
// usage: selfplay [--executor hip|random|zero] [--weights file.nsgw] [--gpu 0]
//                 [--threads 2] [--games-per-group 256] [--playouts 800]
//                 [--seconds 30] [--max-games 0] [--seed 0] [--precision 3] [--mate-search 1] [--dfpn-nodes 100000]
//                 [--teacher out.nsgt]   (training records of finished games, teacher.h)
#include "selfplay.h"
#include "teacher.h"

#include <nshogi_engine_amd/infer/cpu.h>
#include <nshogi_engine_amd/infer/hip.h>

#include <chrono>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

using namespace nshogi::engine;

int main(int Argc, char* Argv[]) {
    std::string Executor = "hip", Weights, TeacherPath;
    int Gpu = 0, Threads = 2, Precision = NSG_PRECISION_F16X3;
    double Seconds = 30.0;
    uint64_t MaxGames = 0;
    selfplay::Options Opt;
    for (int I = 1; I + 1 < Argc; I += 2) {
        const std::string K = Argv[I], V = Argv[I + 1];
        if (K == "--executor") Executor = V;
        else if (K == "--weights") Weights = V;
        else if (K == "--teacher") TeacherPath = V;
        else if (K == "--dfpn-nodes") Opt.DfpnNodes = std::stoull(V);
        else if (K == "--gpu") Gpu = std::stoi(V);
        else if (K == "--threads") Threads = std::stoi(V);
        else if (K == "--games-per-group") Opt.GamesPerGroup = std::stoi(V);
        else if (K == "--playouts") Opt.NumPlayouts = std::stoi(V);
        else if (K == "--seconds") Seconds = std::stod(V);
        else if (K == "--max-games") MaxGames = std::stoull(V);
        else if (K == "--seed") Opt.Seed = std::stoull(V);
        else if (K == "--precision") Precision = std::stoi(V);
        else if (K == "--full-search-ratio") Opt.FullSearchRatio = std::stod(V);
        else if (K == "--cache-entries") Opt.EvalCacheEntries = std::stoull(V);
        else if (K == "--gumbel") Opt.Gumbel = V != "0";
        else if (K == "--mate-search") Opt.MateSearch = V != "0";
        else if (K == "--num-sampling-moves") Opt.NumSamplingMoves = std::stoi(V);
        else { std::cerr << "unknown option " << K << std::endl; return 2; }
    }
    const bool Hip = Executor == "hip";
    std::vector<std::unique_ptr<infer::Infer>> Execs;
    for (int I = 0; I < Threads * 2; ++I) {
        if (Hip) {
            auto H = std::make_unique<infer::Hip>(Gpu, (uint16_t)Opt.GamesPerGroup, (uint16_t)shogi::NumFeaturePlanes);
            H->setPrecision(Precision);
            H->load(Weights, true);
            Execs.push_back(std::move(H));
        } else if (Executor == "zero") {
            Execs.push_back(std::make_unique<infer::Zero>());
        } else {
            Execs.push_back(std::make_unique<infer::Random>((uint64_t)I)); // one engine state per executor
        }
    }
    std::vector<std::unique_ptr<selfplay::Engine>> Engines;
    for (int T = 0; T < Threads; ++T)
        Engines.push_back(std::make_unique<selfplay::Engine>(Execs[2 * T].get(), Execs[2 * T + 1].get(), Opt,
                                                             (uint64_t)T, Hip));
    std::unique_ptr<selfplay::TeacherWriter> Teacher;
    if (!TeacherPath.empty()) {
        Teacher = std::make_unique<selfplay::TeacherWriter>(TeacherPath);
        for (auto& E : Engines) E->setTeacherWriter(Teacher.get());
    }
    volatile bool Stop = false;
    const auto T0 = std::chrono::steady_clock::now();
    std::vector<std::thread> Workers;
    const uint64_t PerThreadGames = MaxGames ? (MaxGames + Threads - 1) / Threads : 0;
    for (int T = 0; T < Threads; ++T) {
        Workers.emplace_back([&, T]() {
            if (Hip) {
                static_cast<infer::Hip*>(Execs[2 * T].get())->resetGPU();
            }
            Engines[T]->run(&Stop, PerThreadGames);
        });
    }
    if (MaxGames == 0) {
        std::this_thread::sleep_for(std::chrono::duration<double>(Seconds));
        Stop = true;
    }
    for (auto& W : Workers) W.join();
    const double Dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
    selfplay::Stats S;
    uint64_t Digest = 0;
    for (auto& E : Engines) {
        const auto& X = E->stats();
        S.Evaluations += X.Evaluations; S.CacheHits += X.CacheHits; S.Batches += X.Batches;
        S.Playouts += X.Playouts; S.Moves += X.Moves; S.GamesBlack += X.GamesBlack;
        S.GamesWhite += X.GamesWhite; S.GamesDraw += X.GamesDraw;
        S.MovesOfFinishedGames += X.MovesOfFinishedGames; S.MatesFound += X.MatesFound;
        S.TeacherRecords += X.TeacherRecords; S.DfpnMates += X.DfpnMates; S.DfpnNodes += X.DfpnNodes;
        Digest ^= E->moveDigest() * 0x9e3779b97f4a7c15ULL + (uint64_t)(&E - &Engines[0]);
    }
    const double Fin = (double)S.finished();
    std::cout << "{\"executor\": \"" << Executor << "\", \"threads\": " << Threads << ", \"games_per_group\": "
              << Opt.GamesPerGroup << ", \"concurrent_games\": " << Threads * 2 * Opt.GamesPerGroup
              << ", \"playouts_per_move\": " << Opt.NumPlayouts << ", \"seconds\": " << Dt
              << ", \"games_finished\": " << S.finished() << ", \"games_per_sec\": " << Fin / Dt
              << ", \"black\": " << S.GamesBlack << ", \"white\": " << S.GamesWhite << ", \"draw\": " << S.GamesDraw
              << ", \"avg_game_length\": " << (Fin > 0 ? S.MovesOfFinishedGames / Fin : 0.0)
              << ", \"moves\": " << S.Moves << ", \"moves_per_sec\": " << S.Moves / Dt
              << ", \"playouts_per_sec\": " << S.Playouts / Dt << ", \"evals_per_sec\": " << S.Evaluations / Dt
              << ", \"avg_batch\": " << (S.Batches ? (double)S.Evaluations / S.Batches : 0.0)
              << ", \"cache_hit_ratio\": " << (S.Evaluations + S.CacheHits ? (double)S.CacheHits / (S.Evaluations + S.CacheHits) : 0.0)
              << ", \"mate_search\": " << (Opt.MateSearch ? 1 : 0) << ", \"mates_found\": " << S.MatesFound
              << ", \"dfpn_nodes\": " << Opt.DfpnNodes << ", \"dfpn_mates\": " << S.DfpnMates
              << ", \"dfpn_nodes_per_move\": " << (S.Moves ? (double)S.DfpnNodes / S.Moves : 0.0)
              << ", \"teacher_records\": " << S.TeacherRecords
              << ", \"digest\": " << Digest << "}" << std::endl;
    return 0;
}
