// selfplay_main.cc -- self-play driver: G GPUs x T engine threads per GPU, each thread with two
// executors (one per game group).  Role of /root/reference/src/selfplay/main.cc (option names
// follow it where they exist: --num-gpus :33, --num-playouts :43-45, --evaluation-cache-memory-size
// :40-41, --full-search-ratio :54-55, --gumbel :56).  Prints one JSON line: games/sec per the
// reference's definition (finished games / elapsed, saveworker.cc:135-137), average batch size and
// cache-hit ratio (selfplayinfo.cc:51-57,72-78), plus playouts/s and a windowed games/sec that leaves
// the cold start out.
//
// Multi-GPU (selfplay/main.cc:189-195: NumGPUs x workers executors, each re-reading the model):
// games shard by slot -- GPU g's threads own the slots [g*T*2*P, (g+1)*T*2*P) -- with no exchange
// between shards.  Evaluation caches (reference: ONE for the process, main.cc:94-97): one per engine
// thread by default, so that a run is reproducible bit for bit whatever the executor; with
// --share-evaluation-cache 1 the T threads of a GPU shard share one (timing then decides which
// thread's evaluation of a position the others reuse: identical results only if the executor is a
// pure function of the position).  The model file is read ONCE: the first
// executor loads it, every other executor on that GPU shares its packed weights, and the first
// executor of every further GPU takes a peer copy over xGMI (nsg_load_shared).
//
// usage: selfplay [--executor hip|random|zero] [--weights model.onnx|file.nsgw] [--gpu 0] [--num-gpus 1]
//                 [--threads 2] [--workers 1] [--games-per-group 256] [--playouts 800]
//                 (--threads: engines per GPU, each with its own two batches in flight; --workers: host
//                  threads per engine that advance its games between two batches; --solver-threads: threads per
//                  engine that run the df-pn call of judge off the search path, 0 = inline)
//                 [--seconds 30] [--max-games 0] [--seed 0] [--precision 3] [--mate-search 1] [--dfpn-nodes 100000]
//                 [--evaluation-cache-memory-size 1024]   (MB per GPU shard, split over its caches; 0 = no cache)
//                 [--share-evaluation-cache 0]
//                 [--numa 0]   (bind GPU shard g's threads to NUMA node g % nodes and first-touch their pinned
//                               batch buffers there: evaluate::Evaluator, role of evaluator.cc:46-76)
//                 [--teacher out.nsgt]   (training records of finished games, teacher.h)
//                 [--game-log games.txt] (one line per finished game: id winner plies digest moves)
//                 [--leaf-log leaves.txt] (tests: every leaf of every batch -- batch no., slot, SFEN, StateConfig, the
//                                          1376 bytes the engine packed into that slot)
//                 --executor hash: test executor whose outputs are a checksum of the position's own bitboards
#include "selfplay.h"
#include "teacher.h"

#include <nshogi_engine_amd/evaluate/evaluator.h>
#include <nshogi_engine_amd/infer/cpu.h>
#include <nshogi_engine_amd/infer/hip.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <thread>
#include <vector>

using namespace nshogi::engine;

namespace {
// Test executor (--executor hash): every output is a pure function of the position's OWN feature
// bitboards -- logits and rates derived from a checksum of its 1376 bytes -- so a leaf that received
// another slot's outputs, or a batch packed out of order, changes the games; with it the
// grouping-independence tests check the routing of evaluate -> scatter, not just the search.
class HashInfer : public infer::Infer {
 public:
    void computeNonBlocking(const nshogi::ml::FeatureBitboard* Features, std::size_t BatchSize, float* DstPolicy,
                            float* DstWinRate, float* DstDrawRate) override {
        for (std::size_t B = 0; B < BatchSize; ++B) {
            const uint64_t* W = reinterpret_cast<const uint64_t*>(Features + B * shogi::NumFeaturePlanes);
            uint64_t H = 0x9e3779b97f4a7c15ULL;
            for (int I = 0; I < 2 * shogi::NumFeaturePlanes; ++I) H = (H ^ W[I]) * 0xff51afd7ed558ccdULL + (H >> 29);
            for (int I = 0; I < shogi::MoveIndexMax; ++I) {
                uint64_t X = H + (uint64_t)I * 0xc4ceb9fe1a85ec53ULL;
                X ^= X >> 31; X *= 0x9e3779b97f4a7c15ULL; X ^= X >> 29;
                DstPolicy[B * shogi::MoveIndexMax + I] = (float)(X >> 40) * (4.0f / 16777216.0f) - 2.0f;
            }
            DstWinRate[B] = (float)((H >> 20) & 0xffff) / 65535.0f;
            DstDrawRate[B] = (float)((H >> 40) & 0xffff) / 65535.0f * 0.5f;
        }
    }
    void computeBlocking(const nshogi::ml::FeatureBitboard* F, std::size_t N, float* P, float* W, float* D) override {
        computeNonBlocking(F, N, P, W, D);
    }
    void await() override {}
    bool isComputing() override { return false; }
};
} // namespace

int main(int Argc, char* Argv[]) {
    std::string Executor = "hip", Weights, TeacherPath, GameLogPath, LeafLogPath;
    int Gpu = 0, NumGpus = 1, Threads = 2, Precision = NSG_PRECISION_F16X3;
    std::vector<int> GpuMap;
    double Seconds = 30.0;
    uint64_t MaxGames = 0;
    std::size_t CacheMB = 1024; // selfplay/main.cc:40-41
    bool ShareCache = false;
    bool Numa = false;
    selfplay::Options Opt;
    for (int I = 1; I + 1 < Argc; I += 2) {
        const std::string K = Argv[I], V = Argv[I + 1];
        if (K == "--executor") Executor = V;
        else if (K == "--weights" || K == "--model") Weights = V;
        else if (K == "--teacher") TeacherPath = V;
        else if (K == "--game-log") GameLogPath = V;
        else if (K == "--leaf-log") LeafLogPath = V;
        else if (K == "--dfpn-nodes") Opt.DfpnNodes = std::stoull(V);
        else if (K == "--gpu") Gpu = std::stoi(V);
        else if (K == "--num-gpus") NumGpus = std::stoi(V);
        else if (K == "--gpu-map") { // physical device of every GPU shard, e.g. "0,0": two shards on one device
            GpuMap.clear();
            std::size_t P = 0;
            while (P <= V.size()) {
                const std::size_t Q = V.find(',', P);
                GpuMap.push_back(std::stoi(V.substr(P, Q == std::string::npos ? std::string::npos : Q - P)));
                if (Q == std::string::npos) break;
                P = Q + 1;
            }
        }
        else if (K == "--threads") Threads = std::stoi(V);
        else if (K == "--workers" || K == "--num-search-workers") Opt.Workers = std::stoi(V);
        else if (K == "--solver-threads") Opt.SolverThreads = std::stoi(V);
        else if (K == "--games-per-group") Opt.GamesPerGroup = std::stoi(V);
        else if (K == "--playouts" || K == "--num-playouts") Opt.NumPlayouts = std::stoi(V);
        else if (K == "--seconds") Seconds = std::stod(V);
        else if (K == "--max-games") MaxGames = std::stoull(V);
        else if (K == "--seed") Opt.Seed = std::stoull(V);
        else if (K == "--precision") Precision = std::stoi(V);
        else if (K == "--full-search-ratio") Opt.FullSearchRatio = std::stod(V);
        else if (K == "--evaluation-cache-memory-size") CacheMB = (std::size_t)std::stoull(V);
        else if (K == "--share-evaluation-cache") ShareCache = V != "0";
        else if (K == "--numa") Numa = V != "0";
        else if (K == "--gumbel") Opt.Gumbel = V != "0";
        else if (K == "--mate-search") Opt.MateSearch = V != "0";
        else if (K == "--num-sampling-moves") Opt.NumSamplingMoves = std::stoi(V);
        else { std::cerr << "unknown option " << K << std::endl; return 2; }
    }
    if (!GpuMap.empty() && (int)GpuMap.size() != NumGpus) { std::cerr << "--gpu-map needs one device per --num-gpus shard" << std::endl; return 2; }
    if (NumGpus < 1 || Threads < 1 || Opt.GamesPerGroup < 1) { std::cerr << "bad --num-gpus/--threads/--games-per-group" << std::endl; return 2; }
    const bool Hip = Executor == "hip";
    const int NumEngines = NumGpus * Threads;
    Opt.TotalSlots = (uint64_t)NumEngines * 2 * (uint64_t)Opt.GamesPerGroup;
    // executors: engine e = GPU (e / Threads), thread (e % Threads); two executors per engine
    std::vector<std::unique_ptr<infer::Infer>> Execs;
    infer::Hip* FirstOfAll = nullptr;
    infer::Hip* FirstOfGpu = nullptr; // first executor of the GPU shard being filled
    for (int E = 0; E < NumEngines; ++E) {
        const int Shard = E / Threads;
        const int Device = GpuMap.empty() ? Gpu + Shard : GpuMap[(std::size_t)Shard];
        for (int G = 0; G < 2; ++G) {
            if (Hip) {
                auto H = std::make_unique<infer::Hip>(Device, (uint16_t)Opt.GamesPerGroup, (uint16_t)shogi::NumFeaturePlanes);
                H->setPrecision(Precision);
                const bool FirstOfShard = E % Threads == 0 && G == 0;
                if (!FirstOfAll) {
                    H->load(Weights, true); // the one read of the model file
                    FirstOfAll = H.get();
                } else {
                    // first executor of a further GPU shard: peer copy from the very first executor;
                    // every other executor shares the packed weights of its own shard's first
                    H->loadShared(FirstOfShard ? *FirstOfAll : *FirstOfGpu);
                }
                if (FirstOfShard) FirstOfGpu = H.get();
                Execs.push_back(std::move(H));
            } else if (Executor == "zero") {
                Execs.push_back(std::make_unique<infer::Zero>());
            } else if (Executor == "hash") {
                Execs.push_back(std::make_unique<HashInfer>());
            } else {
                Execs.push_back(std::make_unique<infer::Random>((uint64_t)(2 * E + G))); // one engine state per executor
            }
        }
    }
    // CacheMB per GPU shard: one cache for the shard's threads, or an equal share for each of them
    std::vector<std::unique_ptr<selfplay::EvalCache>> Caches;
    const int CachesPerGpu = ShareCache ? 1 : Threads;
    for (int C = 0; C < NumGpus * CachesPerGpu && CacheMB; ++C)
        Caches.push_back(std::make_unique<selfplay::EvalCache>(std::max<std::size_t>(1, CacheMB / (std::size_t)CachesPerGpu)));
    std::unique_ptr<selfplay::TeacherWriter> Teacher;
    if (!TeacherPath.empty()) Teacher = std::make_unique<selfplay::TeacherWriter>(TeacherPath);
    std::unique_ptr<selfplay::GameLog> Log;
    if (!GameLogPath.empty()) Log = std::make_unique<selfplay::GameLog>(GameLogPath);
    std::unique_ptr<selfplay::LeafLog> Leaves;
    if (!LeafLogPath.empty()) Leaves = std::make_unique<selfplay::LeafLog>(LeafLogPath);
    // Every engine is built on the thread that runs it: with --numa the thread is first bound to its
    // GPU shard's NUMA node, so the engine's pinned batch buffers are first-touched there.
    std::vector<std::unique_ptr<selfplay::Engine>> Engines((std::size_t)NumEngines);
    std::atomic<int> Built{0}, Running{NumEngines};
    std::atomic<bool> Go{false};
    std::atomic<bool> Stop{false};
    std::vector<std::thread> Workers;
    const uint64_t PerEngineGames = MaxGames ? (MaxGames + NumEngines - 1) / NumEngines : 0;
    for (int E = 0; E < NumEngines; ++E) {
        Workers.emplace_back([&, E]() {
            if (Numa) evaluate::Evaluator::bindToNumaNode((std::size_t)(E / Threads));
            if (Hip) static_cast<infer::Hip*>(Execs[2 * E].get())->resetGPU(); // selfplay/evaluationworker.cc:62-67
            Engines[(std::size_t)E] = std::make_unique<selfplay::Engine>(
                Execs[2 * E].get(), Execs[2 * E + 1].get(), Opt, (uint64_t)E, Hip,
                CacheMB ? Caches[ShareCache ? E / Threads : E].get() : nullptr);
            Engines[(std::size_t)E]->setTeacherWriter(Teacher.get());
            Engines[(std::size_t)E]->setGameLog(Log.get());
            Engines[(std::size_t)E]->setLeafLog(Leaves.get());
            ++Built;
            while (!Go.load(std::memory_order_acquire)) std::this_thread::yield();
            Engines[(std::size_t)E]->run(&Stop, PerEngineGames);
            --Running;
        });
    }
    while (Built.load() < NumEngines) std::this_thread::yield();
    const auto T0 = std::chrono::steady_clock::now();
    auto Now = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count(); };
    Go.store(true, std::memory_order_release);
    // timeline of (t, finished games) for the windowed rate; also ends a --seconds run
    struct Sample { double T; uint64_t Finished; };
    std::vector<Sample> Timeline;
    while (Running.load() > 0) {
        std::this_thread::sleep_for(std::chrono::milliseconds(MaxGames ? 20 : 100));
        uint64_t Fin = 0;
        for (auto& E : Engines) Fin += E->publishedFinished();
        const double T = Now();
        Timeline.push_back({T, Fin});
        if (MaxGames == 0 && T >= Seconds) Stop = true;
    }
    for (auto& W : Workers) W.join();
    const double Dt = Now();
    selfplay::Stats S;
    uint64_t Digest = 0;
    std::vector<uint64_t> EvalsPerGpu((std::size_t)NumGpus, 0);
    for (int E = 0; E < NumEngines; ++E) {
        const selfplay::Stats X = Engines[E]->stats();
        S.Evaluations += X.Evaluations; S.CacheHits += X.CacheHits; S.Batches += X.Batches;
        S.Playouts += X.Playouts; S.Moves += X.Moves; S.GamesBlack += X.GamesBlack;
        S.GamesWhite += X.GamesWhite; S.GamesDraw += X.GamesDraw;
        S.MovesOfFinishedGames += X.MovesOfFinishedGames; S.MatesFound += X.MatesFound;
        S.TeacherRecords += X.TeacherRecords; S.DfpnMates += X.DfpnMates; S.DfpnNodes += X.DfpnNodes;
        S.AwaitNs += X.AwaitNs; S.AwaitsIdle += X.AwaitsIdle; S.HostNs += X.HostNs;
        Digest += Engines[E]->moveDigest();
        EvalsPerGpu[(std::size_t)(E / Threads)] += X.Evaluations;
    }
    // games finished in the second half of the run / its length: the cold start (no game can end
    // before ~one game length has been played in every slot) is left out
    double WindowRate = 0.0, WindowSeconds = 0.0;
    for (const Sample& A : Timeline) {
        if (A.T < Dt / 2) continue;
        WindowSeconds = Dt - A.T;
        if (WindowSeconds > 0) WindowRate = (double)(S.finished() - A.Finished) / WindowSeconds;
        break;
    }
    const double Fin = (double)S.finished();
    std::cout << "{\"executor\": \"" << Executor << "\", \"num_gpus\": " << NumGpus << ", \"threads\": " << Threads << ", \"workers\": " << Opt.Workers << ", \"solver_threads\": " << Opt.SolverThreads
              << ", \"games_per_group\": " << Opt.GamesPerGroup << ", \"concurrent_games\": " << Opt.TotalSlots
              << ", \"playouts_per_move\": " << Opt.NumPlayouts << ", \"seconds\": " << Dt
              << ", \"games_finished\": " << S.finished() << ", \"games_per_sec\": " << Fin / Dt
              << ", \"games_per_sec_window\": " << WindowRate << ", \"window_seconds\": " << WindowSeconds
              << ", \"black\": " << S.GamesBlack << ", \"white\": " << S.GamesWhite << ", \"draw\": " << S.GamesDraw
              << ", \"avg_game_length\": " << (Fin > 0 ? S.MovesOfFinishedGames / Fin : 0.0)
              << ", \"moves\": " << S.Moves << ", \"moves_per_sec\": " << S.Moves / Dt
              << ", \"playouts_per_sec\": " << S.Playouts / Dt << ", \"evals_per_sec\": " << S.Evaluations / Dt
              << ", \"avg_batch\": " << (S.Batches ? (double)S.Evaluations / S.Batches : 0.0)
              << ", \"cache_hit_ratio\": " << (S.Evaluations + S.CacheHits ? (double)S.CacheHits / (S.Evaluations + S.CacheHits) : 0.0)
              << ", \"evaluation_cache_mb_per_gpu\": " << CacheMB << ", \"evaluation_cache_shared\": " << (ShareCache ? 1 : 0)
              << ", \"mate_search\": " << (Opt.MateSearch ? 1 : 0) << ", \"mates_found\": " << S.MatesFound
              << ", \"dfpn_nodes\": " << Opt.DfpnNodes << ", \"dfpn_mates\": " << S.DfpnMates
              << ", \"dfpn_nodes_per_move\": " << (S.Moves ? (double)S.DfpnNodes / S.Moves : 0.0)
              << ", \"teacher_records\": " << S.TeacherRecords
              << ", \"await_ms_per_batch\": " << (S.Batches ? S.AwaitNs * 1e-6 / S.Batches : 0.0)
              << ", \"host_ms_per_batch\": " << (S.Batches ? S.HostNs * 1e-6 / S.Batches : 0.0)
              << ", \"batches_found_finished\": " << (S.Batches ? (double)S.AwaitsIdle / S.Batches : 0.0)
              << ", \"evals_per_sec_by_gpu\": [";
    for (int D = 0; D < NumGpus; ++D) std::cout << (D ? ", " : "") << EvalsPerGpu[(std::size_t)D] / Dt;
    std::cout << "], \"digest\": " << Digest << "}" << std::endl;
    return 0;
}
