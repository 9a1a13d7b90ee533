// infer/hip.h -- `infer::Hip`, the MI355X executor, as a drop-in sibling of
// `infer::TensorRT` (/root/reference/src/infer/trt.h:42-88).
//
// Header-only adapter over the C ABI of include/nsg.h: same constructor
// arguments, same `load`, the four `Infer` virtuals and `resetGPU`, and the
// same failure behaviour the callers already rely on:
//   * file cannot be opened / parsed  -> std::runtime_error  (trt.cc:34-36,127-131)
//   * wrong policy width, bind errors -> message on std::cerr + std::abort()
//                                        (trt.cc:205-227)
//   * device/runtime errors           -> message on std::cerr + exit(1)
//                                        (TRTLogger, trt.h:34-39)
// so src/mcts/evaluationworker.cc and src/selfplay/evaluationworker.cc need
// one more #elif in their executor ladders and nothing else.
#ifndef NSG_INFER_HIP_H
#define NSG_INFER_HIP_H

#include "infer.h"
#include "../../nsg.h"

#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>

namespace nshogi {
namespace engine {
namespace infer {

class Hip : public Infer {
 public:
    Hip(int GPUId, uint16_t BatchSizeMax, uint16_t NumChannels)
        : Handle(nullptr) {
        check(nsg_create(GPUId, BatchSizeMax, NumChannels, &Handle));
    }

    ~Hip() override {
        nsg_destroy(Handle);
    }

    Hip(const Hip&) = delete;
    Hip& operator=(const Hip&) = delete;

    // One of NSG_PRECISION_FP32 (exact f32 MFMA; what a new evaluator starts with unless the
    // environment variable NSG_PRECISION names another default), _F16X3 (f32-equivalent split f16),
    // _F16M6 (split f16 with fp6 MX correction terms: 1.5e-4 from a CPU fp32 executor, 5x the f32
    // rate -- the recommended setting and what bench.py measures), _F16M8, _FP16, _BF16 (nsg.h).
    // Plays the role of BuilderFlag::kTF32 (trt.cc:160-161); call before load().
    void setPrecision(int Precision) {
        check(nsg_set_precision(Handle, Precision));
    }

    // The second argument is accepted for signature parity with
    // TensorRT::load (trt.h:47); there is no engine-build step to cache.
    void load(const std::string& Path, bool /*UseSerializedFileIfAvailable*/) {
        const int RC = nsg_load(Handle, Path.c_str());
        if (RC == NSG_E_IO) {
            throw std::runtime_error(nsg_last_error());
        }
        if (RC == NSG_E_FORMAT) {
            const std::string Msg = nsg_last_error();
            if (Msg.rfind("Unexpected PolicySize", 0) == 0) {
                std::cerr << Msg << std::endl;
                std::abort();
            }
            throw std::runtime_error("Could not parse the model: " + Msg);
        }
        check(RC);
    }

    // Extension: adopt the network `Src` has loaded instead of reading the model file again
    // (the reference's G x T executors each call load(), selfplay/main.cc:189-195): executors
    // on one device share one copy of the weights, another device takes a peer copy over xGMI.
    void loadShared(Hip& Src) {
        check(nsg_load_shared(Handle, Src.Handle));
    }

    void computeNonBlocking(const ml::FeatureBitboard* Features,
                            std::size_t BatchSize, float* DstPolicy,
                            float* DstWinRate, float* DstDrawRate) override {
        check(nsg_compute_nonblocking(Handle, Features, BatchSize, DstPolicy,
                                      DstWinRate, DstDrawRate));
    }

    void computeBlocking(const ml::FeatureBitboard* Features,
                         std::size_t BatchSize, float* DstPolicy,
                         float* DstWinRate, float* DstDrawRate) override {
        computeNonBlocking(Features, BatchSize, DstPolicy, DstWinRate,
                           DstDrawRate);
        await();
    }

    // Extension (not part of infer::Infer): the legal-move lookup of
    // feedworker.cc:120-127 / frame.cc:105-118 on the device.  MoveOffsets: BatchSize+1
    // prefix sums; MoveIndices: ml::getMoveIndex values; DstValues receives the logits (or,
    // with Softmax, the priors) of exactly those moves.
    void computeGatherNonBlocking(const ml::FeatureBitboard* Features, std::size_t BatchSize,
                                  const uint16_t* MoveIndices, const uint32_t* MoveOffsets,
                                  bool Softmax, float* DstValues, float* DstWinRate,
                                  float* DstDrawRate) {
        check(nsg_compute_gather_nonblocking(Handle, Features, BatchSize, MoveIndices, MoveOffsets,
                                             Softmax ? 1 : 0, DstValues, DstWinRate, DstDrawRate));
    }

    void await() override {
        check(nsg_await(Handle));
    }

    bool isComputing() override {
        return nsg_is_computing(Handle) != 0;
    }

    void resetGPU() {
        check(nsg_reset_gpu(Handle));
    }

    nsg_evaluator* handle() {
        return Handle;
    }

 private:
    static void check(int RC) {
        if (RC != NSG_OK) {
            std::cerr << nsg_last_error() << std::endl;
            std::exit(1);
        }
    }

    nsg_evaluator* Handle;
};

} // namespace infer
} // namespace engine
} // namespace nshogi

#endif // NSG_INFER_HIP_H
