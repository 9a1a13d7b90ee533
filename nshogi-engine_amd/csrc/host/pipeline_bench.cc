// pipeline_bench.cc -- end-to-end host path: P search-like producer threads ->
// batching -> HIP executor(s) -> F feed threads, in evaluations per second
// INCLUDING feature packing, H2D, D2H and the per-leaf feed work
// (BASELINE config 3: "batch=512, 20x256, 4 MCTS worker threads").
//
// mode "pipeline":  evaluate::BatchPipeline (this repository's a9/a10).
// mode "reference": the engine's own structure restated as the baseline --
//   one mutex + condvar queue of feature stacks (evaluationqueue.cc:45-90),
//   per-item memcpy into the Evaluator buffer, computeNonBlocking/await, six heap
//   allocations + output memcpy per batch (evaluationworker.cc:124-195), a
//   mutex feed queue (feedqueue.h:26-80).
//
// usage: pipeline_bench <weights.nsgw> <mode> [seconds=5] [producers=4] [batch=512]
//                       [depth=2] [feeders=4] [precision=4]
#include <nshogi_engine_amd/evaluate/batchpipeline.h>
#include <nshogi_engine_amd/evaluate/evaluator.h>
#include <nshogi_engine_amd/infer/hip.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <queue>
#include <thread>
#include <vector>

using namespace nshogi;
using namespace nshogi::engine;

namespace {
constexpr std::size_t kC = 86;
constexpr int kLegalMoves = 80; // typical shogi branching factor

uint64_t mix(uint64_t X) {
    X ^= X >> 33; X *= 0xff51afd7ed558ccdULL; X ^= X >> 33; X *= 0xc4ceb9fe1a85ec53ULL; X ^= X >> 33;
    return X;
}

struct Base {
    ml::FeatureBitboard F[kC];
    Base() {
        for (std::size_t C = 0; C < kC; ++C) {
            const uint64_t R = mix(C + 7);
            F[C].Lo = (C < 28) ? (R & mix(R) & mix(R + 3) & ((1ULL << 63) - 1)) : ((R & 3) == 0 ? (1ULL << 63) - 1 : 0);
            F[C].Hi = ((C < 28) ? 0 : ((R & 3) == 0 ? (1ULL << 18) - 1 : 0)) | (0x3f800000ULL << 32);
        }
    }
};

// what a search thread does per leaf: build the position's 1376-byte feature stack
inline void buildFeatures(const Base& B, uint64_t Id, ml::FeatureBitboard* Dst) {
    std::memcpy(static_cast<void*>(Dst), B.F, sizeof(B.F));
    Dst[Id % 28].Lo ^= 1ULL << (Id % 63);
    Dst[(Id >> 6) % 28].Lo ^= 1ULL << ((Id >> 3) % 63);
    if (Id & 1) for (std::size_t C = 0; C < kC; ++C) Dst[C].Hi |= 1ULL << 24;
}

// what a feed thread does per leaf: gather legal-move logits + softmax (feedworker.cc:120-127)
inline float feedWork(uint64_t Hash, const float* Policy, float Win) {
    float Logit[kLegalMoves];
    float Max = -1e30f;
    for (int I = 0; I < kLegalMoves; ++I) {
        Logit[I] = Policy[mix(Hash + I) % ml::MoveIndexMax];
        Max = std::fmax(Max, Logit[I]);
    }
    float Sum = 0.f;
    for (int I = 0; I < kLegalMoves; ++I) { Logit[I] = std::exp(Logit[I] - Max); Sum += Logit[I]; }
    return Logit[0] / Sum + Win;
}
} // namespace

int main(int Argc, char* Argv[]) {
    if (Argc < 3) {
        std::cerr << "usage: pipeline_bench <weights> pipeline|reference [seconds] [producers] [batch] [depth] [feeders] [precision]" << std::endl;
        return 2;
    }
    const std::string Mode = Argv[2];
    const double Seconds = Argc > 3 ? std::stod(Argv[3]) : 5.0;
    const std::size_t Producers = Argc > 4 ? std::stoul(Argv[4]) : 4;
    const std::size_t Batch = Argc > 5 ? std::stoul(Argv[5]) : 512;
    const std::size_t Depth = Argc > 6 ? std::stoul(Argv[6]) : 2;
    const std::size_t Feeders = Argc > 7 ? std::stoul(Argv[7]) : 4;
    const int Precision = Argc > 8 ? std::stoi(Argv[8]) : NSG_PRECISION_F16M8;

    const Base B;
    std::atomic<bool> Stop{false};
    std::atomic<uint64_t> Fed{0};
    std::atomic<uint64_t> NextId{0};
    std::atomic<uint32_t> Sink{0};
    uint64_t Batches = 0, Positions = 0;

    std::vector<std::unique_ptr<infer::Hip>> Hips;
    const std::size_t NumExec = Mode == "pipeline" ? Depth : 1;
    for (std::size_t I = 0; I < NumExec; ++I) {
        Hips.push_back(std::make_unique<infer::Hip>(0, (uint16_t)Batch, (uint16_t)kC));
        Hips.back()->setPrecision(Precision);
        Hips.back()->load(Argv[1], true);
    }

    const auto T0 = std::chrono::steady_clock::now();
    if (Mode == "pipeline") {
        std::vector<infer::Infer*> Exec;
        for (auto& H : Hips) Exec.push_back(H.get());
        auto Feed = [&](const evaluate::LeafTag& Tag, const float* Policy, float Win, float) {
            Sink.fetch_add((uint32_t)(feedWork(Tag.Hash, Policy, Win) > 0.5f), std::memory_order_relaxed);
            Fed.fetch_add(1, std::memory_order_relaxed);
        };
        evaluate::BatchPipeline Pipe(Exec, kC, Batch, Depth + 2, Feeders, Feed, true);
        std::thread Eval([&]() { for (auto& H : Hips) H->resetGPU(); Pipe.run(); });
        std::vector<std::thread> Prod;
        for (std::size_t P = 0; P < Producers; ++P) {
            Prod.emplace_back([&]() {
                while (!Stop.load(std::memory_order_relaxed)) {
                    const uint64_t Id = NextId.fetch_add(1, std::memory_order_relaxed);
                    evaluate::BatchPipeline::Slot S;
                    if (!Pipe.reserve({nullptr, Id, (uint8_t)(Id & 1)}, &S)) return;
                    buildFeatures(B, Id, S.Features);
                    Pipe.commit(S);
                }
            });
        }
        std::this_thread::sleep_for(std::chrono::duration<double>(Seconds));
        Stop.store(true);
        for (auto& T : Prod) T.join();
        Pipe.close();
        Eval.join();
        Batches = Pipe.stats().Batches;
        Positions = Pipe.stats().Positions;
    } else {
        struct Item { uint64_t Hash; std::vector<ml::FeatureBitboard> FS; };
        struct OutBatch { std::size_t N; std::unique_ptr<uint64_t[]> Hashes; std::unique_ptr<float[]> Pol, Win, Draw;
                          std::unique_ptr<uint8_t[]> Colors; std::unique_ptr<void*[]> Nodes; };
        std::mutex QM, FM; std::condition_variable QCV, FCV;
        std::queue<Item> Q; std::queue<std::unique_ptr<OutBatch>> FQ;
        const std::size_t MaxQ = Batch * 64;
        evaluate::Evaluator Ev(0, kC, Batch, Hips[0].get(), true);
        std::vector<std::thread> Prod, Feed;
        for (std::size_t P = 0; P < Producers; ++P) {
            Prod.emplace_back([&]() {
                while (!Stop.load()) {
                    const uint64_t Id = NextId.fetch_add(1);
                    Item It{Id, std::vector<ml::FeatureBitboard>(kC)};
                    buildFeatures(B, Id, It.FS.data()); // FeatureStack built outside the lock (evaluationqueue.cc:47)
                    std::unique_lock<std::mutex> L(QM);
                    QCV.wait(L, [&]() { return Q.size() < MaxQ || Stop.load(); });
                    if (Stop.load()) return;
                    Q.push(std::move(It));
                }
            });
        }
        for (std::size_t F = 0; F < Feeders; ++F) {
            Feed.emplace_back([&]() {
                for (;;) {
                    std::unique_ptr<OutBatch> OB;
                    {
                        std::unique_lock<std::mutex> L(FM);
                        FCV.wait(L, [&]() { return !FQ.empty() || Stop.load(); });
                        if (FQ.empty()) return;
                        OB = std::move(FQ.front()); FQ.pop();
                    }
                    for (std::size_t I = 0; I < OB->N; ++I) {
                        Sink.fetch_add((uint32_t)(feedWork(OB->Hashes[I], OB->Pol.get() + I * ml::MoveIndexMax, OB->Win[I]) > 0.5f));
                        Fed.fetch_add(1);
                    }
                }
            });
        }
        std::thread Eval([&]() {
            Hips[0]->resetGPU();
            std::vector<uint64_t> Pending(Batch);
            while (!Stop.load()) {
                std::size_t N = 0;
                {
                    std::lock_guard<std::mutex> L(QM);
                    while (!Q.empty() && N < Batch) {
                        Item It = std::move(Q.front()); Q.pop();
                        Pending[N] = It.Hash;
                        std::memcpy(static_cast<void*>(Ev.getFeatureBitboards() + N * kC), It.FS.data(), kC * sizeof(ml::FeatureBitboard));
                        ++N;
                    }
                }
                QCV.notify_all();
                if (N == 0) { std::this_thread::yield(); continue; }
                Ev.computeNonBlocking(N);
                auto OB = std::make_unique<OutBatch>();
                OB->N = N;
                OB->Colors = std::make_unique<uint8_t[]>(N); OB->Nodes = std::make_unique<void*[]>(N);
                OB->Hashes = std::make_unique<uint64_t[]>(N); OB->Pol = std::make_unique<float[]>(N * ml::MoveIndexMax);
                OB->Win = std::make_unique<float[]>(N); OB->Draw = std::make_unique<float[]>(N);
                std::memcpy(OB->Hashes.get(), Pending.data(), N * sizeof(uint64_t));
                Ev.await();
                std::memcpy(OB->Pol.get(), Ev.getPolicy(), N * ml::MoveIndexMax * sizeof(float));
                std::memcpy(OB->Win.get(), Ev.getWinRate(), N * sizeof(float));
                std::memcpy(OB->Draw.get(), Ev.getDrawRate(), N * sizeof(float));
                ++Batches; Positions += N;
                { std::lock_guard<std::mutex> L(FM); FQ.push(std::move(OB)); }
                FCV.notify_one();
            }
        });
        std::this_thread::sleep_for(std::chrono::duration<double>(Seconds));
        Stop.store(true);
        QCV.notify_all(); FCV.notify_all();
        for (auto& T : Prod) T.join();
        Eval.join();
        FCV.notify_all();
        for (auto& T : Feed) T.join();
    }
    const double Dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count();
    std::cout << "{\"mode\": \"" << Mode << "\", \"producers\": " << Producers << ", \"batch\": " << Batch
              << ", \"depth\": " << NumExec << ", \"feeders\": " << Feeders << ", \"precision\": " << Precision
              << ", \"seconds\": " << Dt << ", \"fed\": " << Fed.load() << ", \"evals_per_sec\": " << Fed.load() / Dt
              << ", \"batches\": " << Batches << ", \"avg_batch\": " << (Batches ? (double)Positions / Batches : 0.0)
              << ", \"sink\": " << Sink.load() << "}" << std::endl;
    return 0;
}
