// evaluate/evaluator.h -- `evaluate::Evaluator` for the HIP build: owner of the
// caller-side host batch buffers (FeatureBitboards, Policy, WinRate, DrawRate)
// and thin forwarder to the executor.
//
// Same public interface as /root/reference/src/evaluate/evaluator.h:26-86.
// The only behavioural difference from evaluator.cc:31-149 is the pinning call:
// hipHostRegister (through nsg_host_register) where the reference calls
// cudaHostRegister (evaluator.cc:94-105, 110-115).  NUMA placement is the
// reference's own optional libnuma path and is not restated (libnuma is not in
// this image); allocation falls back to std::malloc exactly as
// evaluator.cc:136-138 does when NUMA is off.
#ifndef NSG_EVALUATE_EVALUATOR_H
#define NSG_EVALUATE_EVALUATOR_H

#include "../infer/infer.h"
#include "../../nsg.h"

#include <cstddef>
#include <cstdlib>

namespace nshogi {
namespace engine {
namespace evaluate {

class Evaluator {
 public:
    // PinMemory: page-lock the four buffers (the reference does so when built
    // with CUDA_ENABLED); pass false for the CPU executors.
    Evaluator(std::size_t /*ThreadId*/, std::size_t FeatureSize,
              std::size_t BatchSize, infer::Infer* In, bool PinMemory = true)
        : PInfer(In)
        , MyFeatureSize(FeatureSize)
        , BatchSizeMax(BatchSize)
        , Pinned(false) {
        FeatureBitboards = static_cast<ml::FeatureBitboard*>(
            allocateMemoryByNumaIfAvailable(BatchSizeMax * MyFeatureSize *
                                            sizeof(ml::FeatureBitboard)));
        Policy = static_cast<float*>(allocateMemoryByNumaIfAvailable(
            ml::MoveIndexMax * BatchSizeMax * sizeof(float)));
        WinRate = static_cast<float*>(
            allocateMemoryByNumaIfAvailable(BatchSizeMax * sizeof(float)));
        DrawRate = static_cast<float*>(
            allocateMemoryByNumaIfAvailable(BatchSizeMax * sizeof(float)));
        if (PinMemory) {
            Pinned =
                nsg_host_register(FeatureBitboards,
                                  BatchSizeMax * MyFeatureSize *
                                      sizeof(ml::FeatureBitboard)) == NSG_OK &&
                nsg_host_register(Policy, ml::MoveIndexMax * BatchSizeMax *
                                              sizeof(float)) == NSG_OK &&
                nsg_host_register(WinRate, BatchSizeMax * sizeof(float)) ==
                    NSG_OK &&
                nsg_host_register(DrawRate, BatchSizeMax * sizeof(float)) ==
                    NSG_OK;
        }
    }

    ~Evaluator() {
        if (Pinned) {
            nsg_host_unregister(FeatureBitboards);
            nsg_host_unregister(Policy);
            nsg_host_unregister(WinRate);
            nsg_host_unregister(DrawRate);
        }
        freeMemory(reinterpret_cast<void**>(&FeatureBitboards), 0);
        freeMemory(reinterpret_cast<void**>(&Policy), 0);
        freeMemory(reinterpret_cast<void**>(&WinRate), 0);
        freeMemory(reinterpret_cast<void**>(&DrawRate), 0);
    }

    Evaluator(const Evaluator&) = delete;
    Evaluator& operator=(const Evaluator&) = delete;

    void computeNonBlocking(std::size_t BatchSize) {
        PInfer->computeNonBlocking(FeatureBitboards, BatchSize, Policy, WinRate,
                                   DrawRate);
    }

    void computeBlocking(std::size_t BatchSize) {
        PInfer->computeBlocking(FeatureBitboards, BatchSize, Policy, WinRate,
                                DrawRate);
    }

    void await() {
        PInfer->await();
    }

    bool isComputing() {
        return PInfer->isComputing();
    }

    inline ml::FeatureBitboard* getFeatureBitboards() {
        return FeatureBitboards;
    }

    inline const float* getPolicy() const {
        return Policy;
    }

    inline const float* getWinRate() const {
        return WinRate;
    }

    inline const float* getDrawRate() const {
        return DrawRate;
    }

    infer::Infer* getInfer() {
        return PInfer;
    }

    bool isPinned() const {
        return Pinned;
    }

    void* allocateMemoryByNumaIfAvailable(std::size_t Size) const {
        // 64-byte alignment: ml::FeatureBitboard needs 16, DMA likes cache lines
        void* Memory = nullptr;
        if (posix_memalign(&Memory, 64, Size == 0 ? 64 : Size) != 0) {
            return nullptr;
        }
        return Memory;
    }

    void freeMemory(void** Memory, std::size_t /*Size*/) const {
        std::free(*Memory);
        *Memory = nullptr;
    }

 private:
    ml::FeatureBitboard* FeatureBitboards;
    float* Policy;
    float* WinRate;
    float* DrawRate;

    infer::Infer* const PInfer;
    const std::size_t MyFeatureSize;
    const std::size_t BatchSizeMax;
    bool Pinned;
};

} // namespace evaluate
} // namespace engine
} // namespace nshogi

#endif // NSG_EVALUATE_EVALUATOR_H
