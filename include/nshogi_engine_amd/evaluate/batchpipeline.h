// evaluate/batchpipeline.h -- host batch-packing pipeline for the HIP executor
// (SURVEY.md 8a a9/a10, 8f #1).
//
// It plays the combined role of the engine's
//   mcts::EvaluationQueue   (/root/reference/src/mcts/evaluationqueue.cc:45-90)
//   mcts::EvaluationWorker  (src/mcts/evaluationworker.cc:105-199)
//   mcts::FeedQueue / Batch (src/mcts/feedqueue.h:26-80)
// with the costs SURVEY.md lists for them removed:
//   * no std::queue of 1.4-KB tuples under one mutex and no per-item memcpy:
//     a search thread reserves a slot with one atomic add and writes the
//     position's feature stack (FeatureType::constructAt) straight into the
//     pinned batch buffer the H2D copy reads from;
//   * no six heap allocations + output memcpy per batch: batches cycle through
//     a fixed pool of pinned buffers that also hold the outputs and the leaf tags;
//   * no blocking bracket around the GPU call: `Depth` executors (one stream each)
//     keep one batch computing while the next fills and the previous is fed back.
//
// Threading: any number of producer threads call reserve()/commit(); ONE
// evaluation thread calls run(); finished batches are delivered to `NumFeeders`
// internal feed threads that invoke the user's FeedFn per leaf (the role of
// FeedWorker::feedResult).  Lock-light: the hot producer path is one fetch_add and
// one release increment; mutex/condvar are used only to sleep when idle or full.
#ifndef NSG_EVALUATE_BATCHPIPELINE_H
#define NSG_EVALUATE_BATCHPIPELINE_H

#include "../infer/infer.h"
#include "../../nsg.h"

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <new>
#include <stdexcept>
#include <thread>
#include <vector>

namespace nshogi {
namespace engine {
namespace evaluate {

// What the engine keeps per queued leaf (evaluationqueue.cc:54-55).
struct LeafTag {
    void* Node;
    uint64_t Hash;
    uint8_t SideToMove;
};

class BatchPipeline {
 public:
    struct Slot {
        ml::FeatureBitboard* Features; // FeatureSize bitboards to fill in place
        uint32_t Buffer;
        uint32_t Index;
    };

    // Called on a feed thread for every evaluated leaf.
    using FeedFn = std::function<void(const LeafTag& Tag, const float* Policy,
                                      float WinRate, float DrawRate)>;

    struct Stats {
        uint64_t Batches = 0;
        uint64_t Positions = 0; // average batch size = Positions / Batches (statistics.h:74-98)
    };

    // Executors: one per in-flight batch (Depth = Executors.size()); they are
    // not owned.  NumBuffers >= Depth + 1.
    BatchPipeline(std::vector<infer::Infer*> Executors, std::size_t FeatureSize,
                  std::size_t BatchSize, std::size_t NumBuffers,
                  std::size_t NumFeeders, FeedFn Feed, bool PinMemory)
        : Exec(std::move(Executors))
        , FeatSize(FeatureSize)
        , BatchMax(BatchSize)
        , Feed(std::move(Feed))
        , Pinned(PinMemory)
        , Buffers(NumBuffers)
        , FillSeq(0)
        , Closed(false) {
        // one buffer filling while every executor has one in flight; fewer deadlocks openNext()
        if (Exec.empty() || NumBuffers < Exec.size() + 1) {
            throw std::invalid_argument("BatchPipeline: NumBuffers must be at least Executors.size() + 1");
        }
        for (auto& B : Buffers) {
            B.Features = static_cast<ml::FeatureBitboard*>(
                alloc(BatchMax * FeatSize * sizeof(ml::FeatureBitboard)));
            B.Policy = static_cast<float*>(
                alloc(BatchMax * ml::MoveIndexMax * sizeof(float)));
            B.Win = static_cast<float*>(alloc(BatchMax * sizeof(float)));
            B.Draw = static_cast<float*>(alloc(BatchMax * sizeof(float)));
            B.Tags.resize(BatchMax);
            B.State.store(0);
            B.Committed.store(0);
            B.Phase.store(kFree);
        }
        Buffers[0].Phase.store(kFilling);
        for (std::size_t I = 0; I < NumFeeders; ++I) {
            Feeders.emplace_back([this]() { feedLoop(); });
        }
    }

    ~BatchPipeline() {
        close();
        stopFeeders();
        for (auto& T : Feeders) {
            T.join();
        }
        for (auto& B : Buffers) {
            release(B.Features);
            release(B.Policy);
            release(B.Win);
            release(B.Draw);
        }
    }

    BatchPipeline(const BatchPipeline&) = delete;
    BatchPipeline& operator=(const BatchPipeline&) = delete;

    // ---- producer side (search threads): EvaluationQueue::add ----------
    // Blocks while every buffer is in flight (the role of MaxQueueSize,
    // evaluationqueue.cc:51).  Returns false once the pipeline is closed.
    bool reserve(const LeafTag& Tag, Slot* Out) {
        for (;;) {
            if (Closed.load(std::memory_order_acquire)) {
                return false;
            }
            const uint64_t Seq = FillSeq.load(std::memory_order_acquire);
            Buffer& B = Buffers[Seq % Buffers.size()];
            if (B.Phase.load(std::memory_order_acquire) == kFilling &&
                B.Seq.load(std::memory_order_acquire) == Seq) {
                // The generation lives in the State word and the slot is taken by compare-exchange,
                // so a producer that read FillSeq one buffer cycle ago can never take (or even
                // count) a slot of the buffer's next generation.
                uint64_t Old = B.State.load(std::memory_order_acquire);
                bool Taken = false;
                while ((Old >> kGenShift) == (Seq & kGenMask) && !(Old & kSealed) && (Old & kCountMask) < BatchMax) {
                    if (B.State.compare_exchange_weak(Old, Old + 1, std::memory_order_acq_rel, std::memory_order_acquire)) {
                        Taken = true;
                        break;
                    }
                }
                if (Taken) {
                    const uint32_t Idx = (uint32_t)(Old & kCountMask);
                    B.Tags[Idx] = Tag;
                    Out->Features = B.Features + (std::size_t)Idx * FeatSize;
                    Out->Buffer = (uint32_t)(Seq % Buffers.size());
                    Out->Index = Idx;
                    return true;
                }
                // full, sealed or recycled: wake the evaluation thread and wait for the next buffer
                wakeEvaluator();
            }
            std::unique_lock<std::mutex> Lock(Mutex);
            ProducerCV.wait_for(Lock, std::chrono::microseconds(50));
        }
    }

    void commit(const Slot& S) {
        Buffer& B = Buffers[S.Buffer];
        const uint64_t C = B.Committed.fetch_add(1, std::memory_order_acq_rel) + 1;
        if (C == 1 || C == BatchMax) {
            wakeEvaluator(); // first leaf of a batch / batch full
        }
    }

    // ---- evaluation thread: EvaluationWorker::doTask -------------------
    // Returns when close() has been called and everything in flight was fed.
    void run() {
        std::deque<InFlight> Flying;
        std::size_t NextExec = 0;
        for (;;) {
            Buffer& B = Buffers[FillSeq.load() % Buffers.size()];
            const bool Closing = Closed.load(std::memory_order_acquire);
            const uint64_t Reserved = B.State.load(std::memory_order_acquire) & kCountMask;
            const bool Full = Reserved >= BatchMax;
            const bool HaveWork = B.Committed.load(std::memory_order_acquire) > 0;

            // retire finished batches first (oldest first), without blocking
            while (!Flying.empty() && !Exec[Flying.front().Exec]->isComputing()) {
                retire(Flying.front());
                Flying.pop_front();
            }

            // launch when a batch is full, or when there is work and an executor
            // idles (the engine takes whatever is queued, evaluationworker.cc:124-154)
            if ((Full || (HaveWork && Flying.size() < Exec.size())) && !Closing) {
                if (Flying.size() == Exec.size()) { // every executor busy: wait for the oldest
                    retire(Flying.front());
                    Flying.pop_front();
                }
                launch(B, NextExec, &Flying);
                NextExec = (NextExec + 1) % Exec.size();
                continue;
            }
            if (Closing) {
                if (HaveWork) { // flush the partial batch
                    if (Flying.size() == Exec.size()) {
                        retire(Flying.front());
                        Flying.pop_front();
                    }
                    launch(B, NextExec, &Flying);
                    NextExec = (NextExec + 1) % Exec.size();
                    continue;
                }
                while (!Flying.empty()) {
                    retire(Flying.front());
                    Flying.pop_front();
                }
                stopFeeders(); // after the last batch has been handed over
                return;
            }
            if (!Flying.empty()) {
                // nothing to launch: block on the oldest batch instead of spinning
                retire(Flying.front());
                Flying.pop_front();
                continue;
            }
            std::unique_lock<std::mutex> Lock(Mutex);
            EvalCV.wait_for(Lock, std::chrono::microseconds(200));
        }
    }

    void close() {
        Closed.store(true, std::memory_order_release);
        {
            std::lock_guard<std::mutex> Lock(Mutex);
        }
        ProducerCV.notify_all();
        EvalCV.notify_all();
    }

    Stats stats() const {
        Stats S;
        S.Batches = NumBatches.load();
        S.Positions = NumPositions.load();
        return S;
    }

 private:
    enum : uint32_t { kFree = 0, kFilling = 1, kInFlight = 2, kFeeding = 3 };
    static constexpr uint64_t kSealed = 1ULL << 32;
    static constexpr uint64_t kCountMask = 0xffffffffULL;
    static constexpr int kGenShift = 33; // State = generation (FillSeq, 31 bits) | sealed | slots handed out
    static constexpr uint64_t kGenMask = 0x7fffffffULL;

    struct Buffer {
        ml::FeatureBitboard* Features = nullptr;
        float* Policy = nullptr;
        float* Win = nullptr;
        float* Draw = nullptr;
        std::vector<LeafTag> Tags;
        std::atomic<uint64_t> State{0};     // low 32: slots handed out; bit 32: sealed; above: generation
        std::atomic<uint64_t> Committed{0}; // slots completely written
        std::atomic<uint32_t> Phase{kFree};
        std::atomic<uint64_t> Seq{0};       // FillSeq value this buffer currently serves
        std::atomic<uint32_t> FeedLeft{0};  // feed ranges not yet finished
        std::size_t Count = 0;
    };

    struct InFlight {
        std::size_t Buf;
        std::size_t Exec;
    };

    struct FeedJob {
        std::size_t Buf;
        std::size_t Begin;
        std::size_t End;
    };

    // Feed threads drain every queued job before they exit.
    void stopFeeders() {
        {
            std::lock_guard<std::mutex> Lock(FeedMutex);
            FeedClosed = true;
        }
        FeedCV.notify_all();
    }

    void wakeEvaluator() {
        EvalCV.notify_one();
    }

    void launch(Buffer& B, std::size_t E, std::deque<InFlight>* Flying) {
        // seal: no slot is handed out after this; N = slots handed out before
        const uint64_t Old = B.State.fetch_or(kSealed, std::memory_order_acq_rel);
        std::size_t N = (std::size_t)(Old & kCountMask);
        if (N > BatchMax) {
            N = BatchMax;
        }
        // open the next buffer for the producers before waiting for stragglers
        const std::size_t Cur = (std::size_t)(FillSeq.load() % Buffers.size());
        openNext();
        while (B.Committed.load(std::memory_order_acquire) < N) {
            std::this_thread::yield(); // a producer is still writing its slot
        }
        B.Count = N;
        B.Phase.store(kInFlight, std::memory_order_release);
        NumBatches.fetch_add(1);
        NumPositions.fetch_add(N);
        Exec[E]->computeNonBlocking(B.Features, N, B.Policy, B.Win, B.Draw);
        Flying->push_back({Cur, E});
    }

    void openNext() {
        const uint64_t Next = FillSeq.load() + 1;
        Buffer& NB = Buffers[Next % Buffers.size()];
        // the pool is sized so that this is normally free already
        while (NB.Phase.load(std::memory_order_acquire) != kFree) {
            std::this_thread::yield();
        }
        // Committed first: a producer can commit only after taking a slot of the new generation,
        // which it can see only through the release store of State below
        NB.Committed.store(0, std::memory_order_relaxed);
        NB.State.store((Next & kGenMask) << kGenShift, std::memory_order_release);
        NB.Seq.store(Next, std::memory_order_release);
        NB.Phase.store(kFilling, std::memory_order_release);
        FillSeq.store(Next, std::memory_order_release);
        ProducerCV.notify_all();
    }

    void retire(const InFlight& F) {
        Exec[F.Exec]->await();
        Buffer& B = Buffers[F.Buf];
        if (Feeders.empty() || B.Count == 0) {
            for (std::size_t I = 0; I < B.Count; ++I) {
                Feed(B.Tags[I], B.Policy + I * ml::MoveIndexMax, B.Win[I], B.Draw[I]);
            }
            B.Phase.store(kFree, std::memory_order_release);
            return;
        }
        // split the batch over the feed threads (FeedWorker::doTask, feedworker.cc:29-53)
        const std::size_t Parts = std::min<std::size_t>(Feeders.size(), B.Count);
        B.FeedLeft.store((uint32_t)Parts, std::memory_order_release);
        B.Phase.store(kFeeding, std::memory_order_release);
        {
            std::lock_guard<std::mutex> Lock(FeedMutex);
            for (std::size_t P = 0; P < Parts; ++P) {
                FeedJobs.push_back({F.Buf, B.Count * P / Parts, B.Count * (P + 1) / Parts});
            }
        }
        FeedCV.notify_all();
    }

    void feedLoop() {
        for (;;) {
            FeedJob J;
            {
                std::unique_lock<std::mutex> Lock(FeedMutex);
                FeedCV.wait(Lock, [this]() { return !FeedJobs.empty() || FeedClosed; });
                if (FeedJobs.empty()) {
                    return;
                }
                J = FeedJobs.front();
                FeedJobs.pop_front();
            }
            Buffer& B = Buffers[J.Buf];
            for (std::size_t I = J.Begin; I < J.End; ++I) {
                Feed(B.Tags[I], B.Policy + I * ml::MoveIndexMax, B.Win[I], B.Draw[I]);
            }
            if (B.FeedLeft.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                B.Phase.store(kFree, std::memory_order_release);
            }
        }
    }

    void* alloc(std::size_t Bytes) {
        void* P = nullptr;
        if (posix_memalign(&P, 4096, Bytes == 0 ? 4096 : Bytes) != 0) {
            throw std::bad_alloc();
        }
        std::memset(P, 0, Bytes);
        // evaluator.cc:94-105; only what was really page-locked is unregistered later
        if (Pinned && nsg_host_register(P, Bytes) == NSG_OK) {
            Registered.push_back(P);
        }
        return P;
    }

    void release(void* P) {
        if (!P) {
            return;
        }
        for (std::size_t I = 0; I < Registered.size(); ++I) {
            if (Registered[I] == P) {
                nsg_host_unregister(P);
                Registered.erase(Registered.begin() + (std::ptrdiff_t)I);
                break;
            }
        }
        std::free(P);
    }

    std::vector<infer::Infer*> Exec;
    const std::size_t FeatSize;
    const std::size_t BatchMax;
    FeedFn Feed;
    const bool Pinned;
    std::vector<void*> Registered; // buffers nsg_host_register accepted

    std::vector<Buffer> Buffers;
    std::atomic<uint64_t> FillSeq;
    std::atomic<bool> Closed;

    std::mutex Mutex;
    std::condition_variable ProducerCV;
    std::condition_variable EvalCV;

    std::vector<std::thread> Feeders;
    std::mutex FeedMutex;
    std::condition_variable FeedCV;
    std::deque<FeedJob> FeedJobs;
    bool FeedClosed = false;

    std::atomic<uint64_t> NumBatches{0};
    std::atomic<uint64_t> NumPositions{0};
};

} // namespace evaluate
} // namespace engine
} // namespace nshogi

#endif // NSG_EVALUATE_BATCHPIPELINE_H
