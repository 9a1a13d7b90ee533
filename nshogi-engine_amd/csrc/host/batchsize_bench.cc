// batchsize_bench.cc -- the reference's own evals/sec benchmark
// (/root/reference/src/bench/batchsize.cc:32-82) restated for the HIP executor:
// identical features in every slot, 4 warm-ups, `Repeat` timed
// Evaluator::computeBlocking(BatchSize) calls (H2D + planes + net + D2H + sync),
// prints "BatchSize, ms, evals/s".
//
// usage: batchsize_bench <weights.nsgw> [repeat=1000] [first=60] [last=159]
//                        [precision=0|1|2] [executor=hip|zero|nothing|random]
// The feature planes replicated across the batch are synthetic (seeded): the
// start position's planes need libnshogi, which is not available here.
#include <nshogi_engine_amd/evaluate/evaluator.h>
#include <nshogi_engine_amd/infer/cpu.h>
#include <nshogi_engine_amd/infer/hip.h>

#include <chrono>
#include <cstring>
#include <iostream>
#include <memory>
#include <random>
#include <string>
#include <vector>

using namespace nshogi;
using namespace nshogi::engine;

namespace {
constexpr std::size_t kFeatureSize = 86; // global_config::FeatureType::size()

std::vector<ml::FeatureBitboard> syntheticPosition() {
    std::mt19937_64 Mt(20240203); // src/test/test_extractbit.cc:72
    std::vector<ml::FeatureBitboard> FS(kFeatureSize);
    for (std::size_t C = 0; C < kFeatureSize; ++C) {
        uint64_t Lo = 0, Hi = 0;
        if (C < 28) { // sparse piece planes
            for (int K = 0; K < 4; ++K) {
                const unsigned Sq = (unsigned)(Mt() % 81);
                if (Sq < 63) Lo |= 1ULL << Sq; else Hi |= 1ULL << (Sq - 63);
            }
        } else if (Mt() % 3 == 0) { // hand / colour planes: all squares or none
            Lo = (1ULL << 63) - 1;
            Hi = (1ULL << 18) - 1;
        }
        Hi |= 0x3f800000ULL << 32; // value 1.0f
        FS[C].Lo = Lo;
        FS[C].Hi = Hi;
    }
    return FS;
}
} // namespace

int main(int Argc, char* Argv[]) {
    if (Argc < 2) {
        std::cerr << "usage: " << Argv[0]
                  << " <weights.nsgw> [repeat] [first] [last] [precision] [executor]" << std::endl;
        return 2;
    }
    const char* WeightPath = Argv[1];
    const std::size_t Repeat = Argc > 2 ? std::stoul(Argv[2]) : 1000;
    const unsigned First = Argc > 3 ? std::stoul(Argv[3]) : 60;
    const unsigned Last = Argc > 4 ? std::stoul(Argv[4]) : 159;
    const int Precision = Argc > 5 ? std::stoi(Argv[5]) : NSG_PRECISION_FP32;
    const std::string Executor = Argc > 6 ? Argv[6] : "hip";

    std::cout << "Bench batch size for the weight file " << WeightPath << " with " << Repeat
              << " repeats." << std::endl;
    const auto FeatureStack = syntheticPosition();

    for (unsigned BatchSize = First; BatchSize <= Last; ++BatchSize) {
        std::unique_ptr<infer::Infer> Infer;
        if (Executor == "hip") {
            auto H = std::make_unique<infer::Hip>(0, (uint16_t)BatchSize, (uint16_t)kFeatureSize);
            H->setPrecision(Precision);
            H->load(WeightPath, false);
            Infer = std::move(H);
        } else if (Executor == "zero") {
            Infer = std::make_unique<infer::Zero>();
        } else if (Executor == "nothing") {
            Infer = std::make_unique<infer::Nothing>();
        } else {
            Infer = std::make_unique<infer::Random>(0);
        }
        evaluate::Evaluator Evaluator(0, kFeatureSize, BatchSize, Infer.get(), Executor == "hip");

        for (std::size_t I = 0; I < BatchSize; ++I) {
            std::memcpy(static_cast<void*>(Evaluator.getFeatureBitboards() + I * kFeatureSize),
                        FeatureStack.data(), kFeatureSize * sizeof(ml::FeatureBitboard));
        }
        for (std::size_t WarmUp = 0; WarmUp < 4; ++WarmUp) {
            Evaluator.computeBlocking(BatchSize);
        }
        const auto StartTime = std::chrono::steady_clock::now();
        for (std::size_t I = 0; I < Repeat; ++I) {
            Evaluator.computeBlocking(BatchSize);
        }
        const auto EndTime = std::chrono::steady_clock::now();
        const double Ms =
            std::chrono::duration_cast<std::chrono::microseconds>(EndTime - StartTime).count() / 1000.0;
        std::cout << BatchSize << ", " << Ms << ", " << (double)(BatchSize * Repeat) / Ms * 1000.0
                  << std::endl;
    }
    return 0;
}
