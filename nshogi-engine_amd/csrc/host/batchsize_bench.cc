// batchsize_bench.cc -- the reference's own evals/sec benchmark
// (/root/reference/src/bench/batchsize.cc:32-82) restated for the HIP executor:
// identical features in every slot, 4 warm-ups, `Repeat` timed
// Evaluator::computeBlocking(BatchSize) calls (H2D + planes + net + D2H + sync),
// prints "BatchSize, ms, evals/s".
//
// usage: batchsize_bench <model.onnx|weights.nsgw> [repeat=1000] [first=60] [last=159]
//                        [precision=0..5|env] [executor=hip|zero|nothing|random]
// precision: an NSG_PRECISION_* number (0 fp32, 1 fp16, 2 bf16, 3 f16x3, 4 f16m8, 5 f16m6) or
// "env" (default): no setPrecision call, exactly what the engine's ladder does -- the evaluator
// then starts with NSG_PRECISION from the environment (nsg.h), fp32 if unset.
// Every slot holds the INITIAL POSITION's 86 planes, like the reference
// (batchsize.cc:47-59: StateBuilder::getInitialState() -> FeatureType), built by this
// repository's rules core (csrc/shogi/features.cc).
#include <nshogi_engine_amd/evaluate/evaluator.h>
#include <nshogi_engine_amd/infer/cpu.h>
#include <nshogi_engine_amd/infer/hip.h>

#include "../shogi/features.h"

#include <chrono>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

using namespace nshogi;
using namespace nshogi::engine;

namespace {
constexpr std::size_t kFeatureSize = shogi::NumFeaturePlanes; // global_config::FeatureType::size()

std::vector<ml::FeatureBitboard> initialPosition() {
    static_assert(sizeof(shogi::FeaturePlane) == sizeof(ml::FeatureBitboard), "one 16-byte layout");
    const shogi::State State; // the initial position
    const shogi::StateConfig Config;
    std::vector<shogi::FeaturePlane> Planes(kFeatureSize);
    shogi::buildFeatures(State, Config, Planes.data());
    std::vector<ml::FeatureBitboard> FS(kFeatureSize);
    std::memcpy(static_cast<void*>(FS.data()), Planes.data(), kFeatureSize * sizeof(ml::FeatureBitboard));
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
    const std::string PrecisionArg = Argc > 5 ? Argv[5] : "env";
    const std::string Executor = Argc > 6 ? Argv[6] : "hip";

    std::cout << "Bench batch size for the weight file " << WeightPath << " with " << Repeat
              << " repeats." << std::endl;
    const auto FeatureStack = initialPosition();

    for (unsigned BatchSize = First; BatchSize <= Last; ++BatchSize) {
        std::unique_ptr<infer::Infer> Infer;
        if (Executor == "hip") {
            auto H = std::make_unique<infer::Hip>(0, (uint16_t)BatchSize, (uint16_t)kFeatureSize);
            if (PrecisionArg != "env") H->setPrecision(std::stoi(PrecisionArg));
            H->load(WeightPath, false);
            if (BatchSize == First) {
                nsg_info Info;
                nsg_get_info(H->handle(), &Info);
                std::cout << "# executor hip on " << Info.device_name << ", precision " << Info.precision
                          << ", " << Info.blocks << "x" << Info.channels << std::endl;
            }
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
