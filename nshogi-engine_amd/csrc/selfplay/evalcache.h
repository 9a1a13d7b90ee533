// evalcache.h -- evaluation cache of the self-play path.
//
// Role and shape of the reference's mcts::EvalCache (/root/reference/src/mcts/evalcache.h:24-67,
// evalcache.cc:16-165; created once in selfplay/main.cc:94-97 and handed to every frame and
// worker, :106,176): position hash -> the legal-move logits + win/draw rates the network
// returned.  Same organisation -- bundles of 3 entries kept in least-recently-used order, bundle
// = Hash % NumBundle, an entry is a hit only if hash AND move count match on store / hash on
// load, at most 164 moves per entry, a busy bundle is skipped rather than waited for
// (evalcache.cc:58-62,133-137: try_lock) -- sized in MB like --evaluation-cache-memory-size.
// Differences: the 3-entry LRU order is a 3-byte permutation instead of a linked list, the lock
// is one atomic flag, and the driver never shares a cache across GPU shards (no
// cross-shard mutex traffic): one per engine thread, or one per GPU with --share-evaluation-cache.
#ifndef NSG_SELFPLAY_EVALCACHE_H
#define NSG_SELFPLAY_EVALCACHE_H

#include <atomic>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>

namespace nshogi {
namespace engine {
namespace selfplay {

class EvalCache {
 public:
    static constexpr std::size_t kMaxMoves = 164; // MAX_CACHE_MOVES_COUNT (evalcache.h:26)
    static constexpr int kBundle = 3;             // CACHE_BUNDLE_SIZE (evalcache.h:47)

    struct Info {
        uint16_t NumMoves;
        float WinRate, DrawRate;
        float Policy[kMaxMoves];
    };

    explicit EvalCache(std::size_t MemoryMB) {
        NumBundle = MemoryMB * 1024ULL * 1024ULL / sizeof(Bundle);
        if (NumBundle == 0) NumBundle = 1;
        // calloc: zero pages are mapped on first touch, so a 1 GiB cache costs what it uses
        Storage = static_cast<Bundle*>(std::calloc(NumBundle, sizeof(Bundle)));
        if (!Storage) throw std::bad_alloc();
        static_assert(std::is_trivially_destructible<Bundle>::value, "bundles live in calloc memory");
    }
    ~EvalCache() { std::free(Storage); }
    EvalCache(const EvalCache&) = delete;
    EvalCache& operator=(const EvalCache&) = delete;

    // evalcache.cc:50-125
    bool store(uint64_t Hash, uint16_t NumMoves, const float* Policy, float WinRate, float DrawRate) {
        if (NumMoves > kMaxMoves) return false;
        Bundle& B = Storage[Hash % NumBundle];
        if (!B.tryLock()) return false;
        int Slot = -1;
        for (int R = 0; R < kBundle; ++R) { // most recent first
            Entry& E = B.Entries[B.order(R)];
            if (!E.Used) { Slot = R; break; }
            if (E.Hash == Hash && E.I.NumMoves == NumMoves) { // already there: refresh its rank only
                B.touch(R);
                B.unlock();
                return true;
            }
            Slot = R; // ends on the least recently used
        }
        Entry& E = B.Entries[B.order(Slot)];
        B.touch(Slot);
        E.Used = 1;
        E.Hash = Hash;
        E.I.NumMoves = NumMoves;
        E.I.WinRate = WinRate;
        E.I.DrawRate = DrawRate;
        std::memcpy(E.I.Policy, Policy, sizeof(float) * NumMoves);
        B.unlock();
        return true;
    }

    // evalcache.cc:127-165; the caller checks NumMoves against the position (worker.cc:370-378)
    bool load(uint64_t Hash, Info* Out) {
        Bundle& B = Storage[Hash % NumBundle];
        if (!B.tryLock()) return false;
        for (int R = 0; R < kBundle; ++R) {
            Entry& E = B.Entries[B.order(R)];
            if (!E.Used) break;
            if (E.Hash == Hash) {
                Out->NumMoves = E.I.NumMoves;
                Out->WinRate = E.I.WinRate;
                Out->DrawRate = E.I.DrawRate;
                std::memcpy(Out->Policy, E.I.Policy, sizeof(float) * E.I.NumMoves);
                B.touch(R);
                B.unlock();
                return true;
            }
        }
        B.unlock();
        return false;
    }

    std::size_t bundles() const { return NumBundle; }

 private:
    struct Entry {
        uint64_t Hash;
        uint8_t Used;
        Info I;
    };
    struct Bundle {
        std::atomic<uint8_t> Lock; // 0 free, 1 held
        uint8_t Rank[kBundle];     // Rank[r] = entry index of the r-th most recently used, stored
                                   // as a delta from the identity so that zeroed memory is valid
        Entry Entries[kBundle];
        bool tryLock() { return Lock.exchange(1, std::memory_order_acquire) == 0; }
        void unlock() { Lock.store(0, std::memory_order_release); }
        int order(int R) const { return (Rank[R] + R) % kBundle; }
        void touch(int R) { // move rank R to the front
            const int E = order(R);
            for (int K = R; K > 0; --K) set(K, order(K - 1));
            set(0, E);
        }
        void set(int R, int EntryIndex) { Rank[R] = (uint8_t)((EntryIndex - R + kBundle) % kBundle); }
    };

    std::size_t NumBundle = 0;
    Bundle* Storage = nullptr;
};

} // namespace selfplay
} // namespace engine
} // namespace nshogi

#endif
