// evaluate/evaluator.h -- `evaluate::Evaluator` for the HIP build: owner of the
// caller-side host batch buffers (FeatureBitboards, Policy, WinRate, DrawRate)
// and thin forwarder to the executor.
//
// Same public interface as /root/reference/src/evaluate/evaluator.h:26-86.
// Differences from evaluator.cc:31-149:
//   * pinning is hipHostRegister (through nsg_host_register) where the reference calls
//     cudaHostRegister (evaluator.cc:94-105, 110-115), tracked per buffer so that exactly
//     what was page-locked is unregistered;
//   * NUMA placement (the reference's optional NUMA_ENABLED build, evaluator.cc:46-76,127-149:
//     node = AvailableNodes[ThreadId % count], sched_setaffinity to the node's CPUs,
//     numa_alloc_onnode) needs no libnuma here: the node list and CPU masks come from
//     /sys/devices/system/node, the thread is bound with sched_setaffinity, and the buffers are
//     first-touched right after, which places their pages on that node under the default local
//     allocation policy.  Off unless asked for (NumaPlacement = true), like the reference's build
//     flag; a machine with one node is left alone.
#ifndef NSG_EVALUATE_EVALUATOR_H
#define NSG_EVALUATE_EVALUATOR_H

#include "../infer/infer.h"
#include "../../nsg.h"
#if __has_include(<nshogi/ml/common.h>)
#include <nshogi/ml/common.h> // ml::MoveIndexMax (as src/evaluate/evaluator.h:19)
#endif

#include <sched.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace nshogi {
namespace engine {
namespace evaluate {

class Evaluator {
 public:
    // PinMemory: page-lock the four buffers (the reference does so when built
    // with CUDA_ENABLED); pass false for the CPU executors.
    Evaluator(std::size_t ThreadId, std::size_t FeatureSize,
              std::size_t BatchSize, infer::Infer* In, bool PinMemory = true,
              bool NumaPlacement = false)
        : PInfer(In)
        , MyFeatureSize(FeatureSize)
        , BatchSizeMax(BatchSize)
        , Pinned(false)
        , MyNumaId(-1) {
        if (NumaPlacement) {
            MyNumaId = bindToNumaNode(ThreadId); // evaluator.cc:46-76
        }
        const std::size_t Sizes[4] = {
            BatchSizeMax * MyFeatureSize * sizeof(ml::FeatureBitboard),
            ml::MoveIndexMax * BatchSizeMax * sizeof(float),
            BatchSizeMax * sizeof(float), BatchSizeMax * sizeof(float)};
        void* Mem[4];
        for (int I = 0; I < 4; ++I) {
            Mem[I] = allocateMemoryByNumaIfAvailable(Sizes[I]);
            if (!Mem[I]) {
                for (int J = 0; J < I; ++J) std::free(Mem[J]);
                throw std::bad_alloc();
            }
            std::memset(Mem[I], 0, Sizes[I]); // first touch: pages land on this thread's node
        }
        FeatureBitboards = static_cast<ml::FeatureBitboard*>(Mem[0]);
        Policy = static_cast<float*>(Mem[1]);
        WinRate = static_cast<float*>(Mem[2]);
        DrawRate = static_cast<float*>(Mem[3]);
        if (PinMemory) {
            Pinned = true;
            for (int I = 0; I < 4; ++I) {
                Registered[I] = nsg_host_register(Mem[I], Sizes[I]) == NSG_OK;
                Pinned = Pinned && Registered[I];
            }
        }
    }

    ~Evaluator() {
        void* Mem[4] = {FeatureBitboards, Policy, WinRate, DrawRate};
        for (int I = 0; I < 4; ++I) {
            if (Registered[I]) nsg_host_unregister(Mem[I]);
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

    // NUMA node this evaluator's thread and buffers were placed on; -1 = no placement.
    int numaNode() const {
        return MyNumaId;
    }

    // CPU lists of the machine's NUMA nodes that have CPUs, from /sys/devices/system/node.
    static std::vector<std::vector<int>> numaNodeCpus() {
        std::vector<std::vector<int>> Nodes;
        for (int Node = 0; Node < 1024; ++Node) {
            const std::string Path = "/sys/devices/system/node/node" + std::to_string(Node) + "/cpulist";
            std::FILE* F = std::fopen(Path.c_str(), "r");
            if (!F) {
                if (Node > 64) break; // node numbers may be sparse, but not that sparse
                continue;
            }
            char Buf[4096] = {0};
            const std::size_t Got = std::fread(Buf, 1, sizeof(Buf) - 1, F);
            std::fclose(F);
            std::vector<int> Cpus; // "0-31,64-95"
            std::size_t P = 0;
            while (P < Got) {
                char* End = nullptr;
                const long A = std::strtol(Buf + P, &End, 10);
                if (End == Buf + P) break;
                long B = A;
                P = (std::size_t)(End - Buf);
                if (Buf[P] == '-') {
                    B = std::strtol(Buf + P + 1, &End, 10);
                    P = (std::size_t)(End - Buf);
                }
                for (long C = A; C <= B; ++C) Cpus.push_back((int)C);
                if (Buf[P] == ',') ++P; else break;
            }
            if (!Cpus.empty()) Nodes.push_back(std::move(Cpus));
        }
        return Nodes;
    }

    // Binds the calling thread to the CPUs of node Nodes[ThreadId % count]; returns the index, or
    // -1 when the machine has a single node or the call fails.
    static int bindToNumaNode(std::size_t ThreadId) {
        const auto Nodes = numaNodeCpus();
        if (Nodes.size() < 2) return -1;
        const std::size_t Mine = ThreadId % Nodes.size();
        cpu_set_t Set;
        CPU_ZERO(&Set);
        for (int C : Nodes[Mine]) {
            if (C < CPU_SETSIZE) CPU_SET(C, &Set);
        }
        return sched_setaffinity(0, sizeof(Set), &Set) == 0 ? (int)Mine : -1;
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
    bool Registered[4] = {false, false, false, false};
    int MyNumaId;
};

} // namespace evaluate
} // namespace engine
} // namespace nshogi

#endif // NSG_EVALUATE_EVALUATOR_H
