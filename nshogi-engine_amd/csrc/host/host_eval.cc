// host_eval.cc -- drives infer::Hip + evaluate::Evaluator the way the engine's
// evaluation thread does (/root/reference/src/mcts/evaluationworker.cc:124-195):
// memcpy feature stacks into the pinned Evaluator buffer, computeNonBlocking,
// do host work, await, copy the outputs out -- over a sequence of batches of
// varying size.  Inputs/outputs are raw binary files so the Python parity tests
// can compare against the oracle.
//
// usage: host_eval <weights.nsgw> <features.bin> <count> <batch_max> <out.bin>
//                  [precision] [executor=hip|zero|nothing|random]
// out.bin = count * (2187 + 2) floats, position-major: policy, win, draw.
#include <nshogi_engine_amd/evaluate/evaluator.h>
#include <nshogi_engine_amd/infer/cpu.h>
#include <nshogi_engine_amd/infer/hip.h>

#include <cstdio>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

using namespace nshogi;
using namespace nshogi::engine;

int main(int Argc, char* Argv[]) {
    if (Argc < 6) {
        std::cerr << "usage: host_eval <weights> <features.bin> <count> <batch_max> <out.bin> "
                     "[precision] [executor]" << std::endl;
        return 2;
    }
    const std::size_t C = 86;
    const std::size_t Count = std::stoul(Argv[3]);
    const std::size_t BatchMax = std::stoul(Argv[4]);
    const int Precision = Argc > 6 ? std::stoi(Argv[6]) : 0;
    const std::string Executor = Argc > 7 ? Argv[7] : "hip";

    std::vector<ml::FeatureBitboard> Features(Count * C);
    FILE* F = std::fopen(Argv[2], "rb");
    if (!F || std::fread(Features.data(), sizeof(ml::FeatureBitboard), Count * C, F) != Count * C) {
        std::cerr << "cannot read " << Argv[2] << std::endl;
        return 2;
    }
    std::fclose(F);

    std::unique_ptr<infer::Infer> Infer;
    try {
        if (Executor == "hip") {
            auto H = std::make_unique<infer::Hip>(0, (uint16_t)BatchMax, (uint16_t)C);
            H->setPrecision(Precision);
            H->load(Argv[1], true);
            H->resetGPU(); // mcts/evaluationworker.cc:85
            Infer = std::move(H);
        } else if (Executor == "zero") {
            Infer = std::make_unique<infer::Zero>();
        } else if (Executor == "nothing") {
            Infer = std::make_unique<infer::Nothing>();
        } else {
            Infer = std::make_unique<infer::Random>(0);
        }
    } catch (const std::runtime_error& E) {
        std::cerr << "runtime_error: " << E.what() << std::endl;
        return 3;
    }
    evaluate::Evaluator Evaluator(0, C, BatchMax, Infer.get(), Executor == "hip");

    std::vector<float> Out(Count * (ml::MoveIndexMax + 2), -1.0f);
    std::size_t Done = 0, Step = 0;
    while (Done < Count) {
        // batch sizes cycle 1, max, 2, max-1, ... like a draining queue would
        std::size_t N = (Step % 2 == 0) ? 1 + (Step / 2) % BatchMax : BatchMax - (Step / 2) % BatchMax;
        if (N > Count - Done) N = Count - Done;
        ++Step;
        for (std::size_t I = 0; I < N; ++I) { // getBatch(): per-item memcpy (evaluationworker.cc:146-150)
            std::memcpy(static_cast<void*>(Evaluator.getFeatureBitboards() + I * C),
                        Features.data() + (Done + I) * C, C * sizeof(ml::FeatureBitboard));
        }
        Evaluator.computeNonBlocking(N); // doInference(): evaluationworker.cc:158
        // ... the engine copies batch metadata here while the GPU runs (:160-174)
        Evaluator.await(); // :180
        for (std::size_t I = 0; I < N; ++I) { // :183-188
            float* Dst = Out.data() + (Done + I) * (ml::MoveIndexMax + 2);
            std::memcpy(Dst, Evaluator.getPolicy() + I * ml::MoveIndexMax, ml::MoveIndexMax * sizeof(float));
            Dst[ml::MoveIndexMax] = Evaluator.getWinRate()[I];
            Dst[ml::MoveIndexMax + 1] = Evaluator.getDrawRate()[I];
        }
        if (Evaluator.isComputing()) {
            std::cerr << "isComputing() true after await()" << std::endl;
            return 4;
        }
        Done += N;
    }
    F = std::fopen(Argv[5], "wb");
    if (!F || std::fwrite(Out.data(), sizeof(float), Out.size(), F) != Out.size()) {
        std::cerr << "cannot write " << Argv[5] << std::endl;
        return 2;
    }
    std::fclose(F);
    std::cout << "ok " << Count << " positions in " << Step << " batches, pinned="
              << Evaluator.isPinned() << std::endl;
    return 0;
}
