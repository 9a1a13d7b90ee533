// infer/cpu.h -- the reference's CPU stand-in executors `infer::Zero`,
// `infer::Nothing` and `infer::Random` (/root/reference/src/infer/zero.cc:25-45,
// nothing.cc:22-35, random.cc:21-57) over the C ABI, so that a build of this
// repository's host code can be driven with EXECUTOR=zero|nothing|random exactly
// like the engine (Makefile:107-121).
#ifndef NSG_INFER_CPU_H
#define NSG_INFER_CPU_H

#include "infer.h"
#include "../../nsg.h"

#include <cstdint>

namespace nshogi {
namespace engine {
namespace infer {

class CpuExecutor : public Infer {
 public:
    ~CpuExecutor() override {
        nsg_cpu_executor_destroy(Handle);
    }
    void computeNonBlocking(const ml::FeatureBitboard* Features,
                            std::size_t BatchSize, float* DstPolicy,
                            float* DstWinRate, float* DstDrawRate) override {
        nsg_cpu_executor_compute(Handle, Features, BatchSize, DstPolicy,
                                 DstWinRate, DstDrawRate);
    }
    void computeBlocking(const ml::FeatureBitboard* Features,
                         std::size_t BatchSize, float* DstPolicy,
                         float* DstWinRate, float* DstDrawRate) override {
        computeNonBlocking(Features, BatchSize, DstPolicy, DstWinRate,
                           DstDrawRate);
        await();
    }
    void await() override {
    }
    bool isComputing() override {
        return false;
    }

 protected:
    CpuExecutor(int Kind, uint64_t Seed)
        : Handle(nullptr) {
        nsg_cpu_executor_create(Kind, Seed, &Handle);
    }

 private:
    nsg_cpu_executor* Handle;
};

class Zero : public CpuExecutor {
 public:
    Zero() : CpuExecutor(0, 0) {
    }
};

class Nothing : public CpuExecutor {
 public:
    Nothing() : CpuExecutor(1, 0) {
    }
};

class Random : public CpuExecutor {
 public:
    explicit Random(uint64_t Seed) : CpuExecutor(2, Seed) {
    }
};

} // namespace infer
} // namespace engine
} // namespace nshogi

#endif // NSG_INFER_CPU_H
