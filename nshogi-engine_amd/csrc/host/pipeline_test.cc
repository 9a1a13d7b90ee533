// pipeline_test.cc -- concurrency / routing test of evaluate::BatchPipeline.
//
// mode "checksum" (no GPU): a fake executor derives every output from the
// position's own feature bytes, so a result delivered to the wrong leaf, lost or
// duplicated is detected.  mode "hip": the same leaves are first evaluated one
// blocking batch at a time through evaluate::Evaluator (the reference's plain
// path, src/mcts/evaluationworker.cc:124-195) and the pipelined results must be
// bit-identical (the f32 path is batch-composition independent).
//
// usage: pipeline_test checksum|hip <leaves> <producers> <batch> <buffers> <depth> <feeders> [weights.nsgw]
#include <nshogi_engine_amd/evaluate/batchpipeline.h>
#include <nshogi_engine_amd/evaluate/evaluator.h>
#include <nshogi_engine_amd/infer/hip.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <thread>
#include <vector>

using namespace nshogi;
using namespace nshogi::engine;

namespace {
constexpr std::size_t kC = 86;

uint64_t mix(uint64_t X) {
    X ^= X >> 33; X *= 0xff51afd7ed558ccdULL; X ^= X >> 33; X *= 0xc4ceb9fe1a85ec53ULL; X ^= X >> 33;
    return X;
}

void makeFeatures(uint64_t Id, ml::FeatureBitboard* F) {
    for (std::size_t C = 0; C < kC; ++C) {
        const uint64_t R = mix(Id * 1315423911ULL + C);
        F[C].Lo = R & ((1ULL << 63) - 1) & mix(R); // sparse-ish
        F[C].Hi = (mix(R + 1) & ((1ULL << 18) - 1)) | ((Id & 1) << 24) | (0x3f800000ULL << 32);
    }
}

float checksum(const ml::FeatureBitboard* F) {
    uint64_t H = 0;
    for (std::size_t C = 0; C < kC; ++C) H = mix(H ^ F[C].Lo) ^ mix(F[C].Hi);
    return (float)(H & 0xfffff);
}

class ChecksumInfer : public infer::Infer {
 public:
    void computeNonBlocking(const ml::FeatureBitboard* Features, std::size_t BatchSize, float* DstPolicy,
                            float* DstWinRate, float* DstDrawRate) override {
        for (std::size_t I = 0; I < BatchSize; ++I) {
            const float S = checksum(Features + I * kC);
            DstPolicy[I * ml::MoveIndexMax] = S;
            DstPolicy[I * ml::MoveIndexMax + ml::MoveIndexMax - 1] = S + 1.0f;
            DstWinRate[I] = S + 2.0f;
            DstDrawRate[I] = S + 3.0f;
        }
    }
    void computeBlocking(const ml::FeatureBitboard* F, std::size_t N, float* P, float* W, float* D) override {
        computeNonBlocking(F, N, P, W, D);
    }
    void await() override {}
    bool isComputing() override { return false; }
};
} // namespace

int main(int Argc, char* Argv[]) {
    if (Argc >= 2 && std::string(Argv[1]) == "numa") {
        // Evaluator's NUMA placement (evaluator.cc:46-76 without libnuma): the machine's nodes, and
        // a few evaluators created with placement on -- each must report the node it was bound to
        const auto Nodes = evaluate::Evaluator::numaNodeCpus();
        std::cout << "nodes " << Nodes.size();
        for (const auto& N : Nodes) std::cout << " " << N.size();
        std::cout << std::endl;
        ChecksumInfer Exec;
        for (std::size_t T = 0; T < 3; ++T) {
            evaluate::Evaluator Ev(T, kC, 4, &Exec, false, true);
            const int Want = Nodes.size() < 2 ? -1 : (int)(T % Nodes.size());
            if (Ev.numaNode() != Want) { std::cout << "wrong node " << Ev.numaNode() << " want " << Want << std::endl; return 1; }
            if (Want >= 0) {
                cpu_set_t Set;
                sched_getaffinity(0, sizeof(Set), &Set);
                for (int C : Nodes[(std::size_t)Want]) if (C < CPU_SETSIZE && !CPU_ISSET(C, &Set)) { std::cout << "cpu not in mask" << std::endl; return 1; }
                if (CPU_COUNT(&Set) != (int)Nodes[(std::size_t)Want].size()) { std::cout << "mask too wide" << std::endl; return 1; }
            }
        }
        std::cout << "numa ok" << std::endl;
        return 0;
    }
    if (Argc >= 2 && std::string(Argv[1]) == "badpool") { // NumBuffers < Depth + 1 must be refused, not deadlock
        ChecksumInfer A, B;
        try {
            evaluate::BatchPipeline P({&A, &B}, kC, 8, 2, 1, [](const evaluate::LeafTag&, const float*, float, float) {}, false);
        } catch (const std::invalid_argument& E) {
            std::cout << "refused: " << E.what() << std::endl;
            return 0;
        }
        return 1;
    }
    if (Argc < 8) {
        std::cerr << "usage: pipeline_test checksum|hip <leaves> <producers> <batch> <buffers> <depth> <feeders> [weights]" << std::endl;
        return 2;
    }
    const std::string Mode = Argv[1];
    const std::size_t Leaves = std::stoul(Argv[2]);
    const std::size_t Producers = std::stoul(Argv[3]);
    const std::size_t Batch = std::stoul(Argv[4]);
    const std::size_t NumBuffers = std::stoul(Argv[5]);
    const std::size_t Depth = std::stoul(Argv[6]);
    const std::size_t Feeders = std::stoul(Argv[7]);
    const bool Hip = Mode == "hip";

    std::vector<std::unique_ptr<infer::Infer>> Owned;
    std::vector<infer::Infer*> Exec;
    for (std::size_t I = 0; I < Depth; ++I) {
        if (Hip) {
            auto H = std::make_unique<infer::Hip>(0, (uint16_t)Batch, (uint16_t)kC);
            H->load(Argv[8], true);
            Exec.push_back(H.get());
            Owned.push_back(std::move(H));
        } else {
            Owned.push_back(std::make_unique<ChecksumInfer>());
            Exec.push_back(Owned.back().get());
        }
    }

    // expected results per leaf id
    std::vector<float> Expect(Leaves * 4, 0.f); // policy[0], policy[last], win, draw
    if (Hip) {
        infer::Hip Ref(0, (uint16_t)Batch, (uint16_t)kC);
        Ref.load(Argv[8], true);
        evaluate::Evaluator Ev(0, kC, Batch, &Ref, true);
        for (std::size_t Done = 0; Done < Leaves;) {
            const std::size_t N = std::min(Batch, Leaves - Done);
            for (std::size_t I = 0; I < N; ++I) makeFeatures(Done + I, Ev.getFeatureBitboards() + I * kC);
            Ev.computeBlocking(N);
            for (std::size_t I = 0; I < N; ++I) {
                float* E = &Expect[(Done + I) * 4];
                E[0] = Ev.getPolicy()[I * ml::MoveIndexMax];
                E[1] = Ev.getPolicy()[I * ml::MoveIndexMax + ml::MoveIndexMax - 1];
                E[2] = Ev.getWinRate()[I];
                E[3] = Ev.getDrawRate()[I];
            }
            Done += N;
        }
    } else {
        std::vector<ml::FeatureBitboard> F(kC);
        for (std::size_t Id = 0; Id < Leaves; ++Id) {
            makeFeatures(Id, F.data());
            const float S = checksum(F.data());
            Expect[Id * 4] = S; Expect[Id * 4 + 1] = S + 1; Expect[Id * 4 + 2] = S + 2; Expect[Id * 4 + 3] = S + 3;
        }
    }

    std::vector<std::atomic<int>> Seen(Leaves);
    for (auto& S : Seen) S.store(0);
    std::atomic<uint64_t> Wrong{0};
    auto Feed = [&](const evaluate::LeafTag& Tag, const float* Policy, float Win, float Draw) {
        const std::size_t Id = (std::size_t)Tag.Hash;
        if (Id >= Leaves || Tag.Node != (void*)(uintptr_t)(Id + 1)) { Wrong.fetch_add(1); return; }
        Seen[Id].fetch_add(1);
        const float* E = &Expect[Id * 4];
        if (std::memcmp(&Policy[0], &E[0], 4) != 0 || std::memcmp(&Policy[ml::MoveIndexMax - 1], &E[1], 4) != 0 ||
            std::memcmp(&Win, &E[2], 4) != 0 || std::memcmp(&Draw, &E[3], 4) != 0) {
            Wrong.fetch_add(1);
        }
    };

    evaluate::BatchPipeline Pipe(Exec, kC, Batch, NumBuffers, Feeders, Feed, Hip);
    std::thread EvalThread([&]() {
        if (Hip) for (auto* E : Exec) static_cast<infer::Hip*>(E)->resetGPU();
        Pipe.run();
    });
    std::atomic<uint64_t> NextId{0};
    std::vector<std::thread> Prod;
    for (std::size_t P = 0; P < Producers; ++P) {
        Prod.emplace_back([&, P]() {
            for (;;) {
                const uint64_t Id = NextId.fetch_add(1);
                if (Id >= Leaves) return;
                evaluate::BatchPipeline::Slot S;
                evaluate::LeafTag Tag{(void*)(uintptr_t)(Id + 1), Id, (uint8_t)(Id & 1)};
                if (!Pipe.reserve(Tag, &S)) return;
                makeFeatures(Id, S.Features); // FeatureType::constructAt straight into the pinned buffer
                Pipe.commit(S);
                if ((Id % 97) == P) std::this_thread::yield();
            }
        });
    }
    for (auto& T : Prod) T.join();
    Pipe.close();
    EvalThread.join();
    // feed threads finish inside ~BatchPipeline; give them time by scoping
    std::size_t Missing = 0, Dup = 0;
    // wait for feeders to drain
    for (int Spin = 0; Spin < 2000; ++Spin) {
        Missing = 0;
        for (auto& S : Seen) Missing += S.load() == 0;
        if (Missing == 0) break;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    for (auto& S : Seen) Dup += S.load() > 1;
    const auto St = Pipe.stats();
    std::cout << "leaves " << Leaves << " batches " << St.Batches << " avg_batch "
              << (St.Batches ? (double)St.Positions / St.Batches : 0.0) << " missing " << Missing << " dup " << Dup
              << " wrong " << Wrong.load() << std::endl;
    return (Missing == 0 && Dup == 0 && Wrong.load() == 0 && St.Positions == Leaves) ? 0 : 1;
}
