// selfplay.h -- self-play rollout engine on top of the batched evaluator
// (BASELINE metric #2 "self-play games/sec").
//
// Follows the reference's AlphaZero-mode frame state machine
// (/root/reference/src/selfplay/worker.cc:55-110 dispatch; prepareRoot :159-215,
// selectLeaf :217-266, checkTerminal :268-381, backpropagate :383-410,
// sequentialHalving (non-Gumbel part) :412-426, transition :528-610, judge
// :477-526; Frame::setEvaluation frame.cc:93-136; PUCT worker.cc:688-715;
// win-rate blending :726-770) with these deliberate differences:
//   * one engine owns its games and steps them in two groups: while one group's
//     leaf batch is on the GPU the engine searches the other group (the reference
//     brackets a blocking GPU call with serial host loops,
//     selfplay/evaluationworker.cc:69-117).  The search between two batches is a fork-join over
//     Options::Workers host threads, each advancing its own fixed share of the group's games
//     (role of --num-search-workers, selfplay/main.cc:34-35), so the batch size stays the group
//     size however many host threads it takes to keep the GPU fed;
//   * every game has its own RNG seeded from (base seed, game id): the driver numbers its
//     concurrent game slots 0..N-1 across all engines and slot s plays the games s, s+N, s+2N, ...
//     Batch results are slot-independent, so a run is reproducible bit for bit, and -- when the
//     executor's arithmetic does not depend on the batch size (CPU executors; the HIP evaluator
//     with a fixed tile plan) -- every game is the same game however the slots are spread over
//     threads, groups and GPUs (the reference seeds from std::random_device, worker.cc:49-50);
//   * search trees live in one bump arena per game, rewound when the root changes (the reference
//     pools nodes and edges in allocator::FixedAllocator / SegregatedFreeListAllocator and frees
//     them on a garbage-collector thread, selfplay/main.cc:72-92);
//   * Gumbel mode (worker.cc:428-475 sequential halving, :596-638 transition,
//     :784-905 sampling / halving schedule) is implemented as in the reference;
//   * the mate-in-3 search at leaves (worker.cc:349-358) is shogi::State::findMate(3) and the
//     df-pn solver call in judge (worker.cc:516-524, 100 000 nodes) is shogi::DfpnSolver
//     (csrc/shogi/dfpn.h), one solver per engine thread;
//   * finished games are replayed into teacher records by TeacherWriter (teacher.h; role of
//     saveworker.cc:160-182) in this build's own record format, libnshogi's SimpleTeacher
//     byte format being absent with the library.
// Rules, feature planes and the policy move index come from csrc/shogi (this
// build's own; parity with libnshogi unpinned).
#ifndef NSG_SELFPLAY_H
#define NSG_SELFPLAY_H

#include "../shogi/dfpn.h"
#include "../shogi/features.h"
#include "../shogi/shogi.h"
#include "evalcache.h"

#include <nshogi_engine_amd/infer/infer.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <memory>
#include <functional>
#include <random>
#include <thread>
#include <vector>

namespace nshogi {
namespace engine {
namespace selfplay {

struct Options {
    int GamesPerGroup = 256;     // concurrent games per group (2 groups per engine)
    int NumPlayouts = 800;       // --num-playouts (selfplay/main.cc:43-45)
    double FullSearchRatio = 0.25; // --full-search-ratio (main.cc:54-55)
    bool Gumbel = false;         // --gumbel: Gumbel AlphaZero root (sequential halving)
    int NumSamplingMoves = 16;   // --num-sampling-moves ("m" of the paper, main.cc:46-48)
    uint64_t Seed = 0;
    int MaxPlyMin = 160 + 64;    // worker.cc:135-136
    int MaxPlyMax = 512 + 128;
    bool RandomDrawValue = true; // worker.cc:142-150
    uint64_t TotalSlots = 0;     // concurrent games of the whole run (all engines); 0 = this engine's own
    int Workers = 1;             // host threads that advance one engine's games between two batches (fork-join)
    // The df-pn call in judge (worker.cc:516-524) can take its whole node budget -- 0.3 s at 100 000 nodes -- and
    // inside a fork-join step that stalls every other game of the engine.  With SolverThreads > 0 a game hands
    // its position to a pool of solver threads and simply contributes no leaf until the answer is there (in
    // the reference the call blocks one of many search workers, and the other frames go on).  The games
    // themselves are unchanged (own RNG, own tree); what changes with timing is which games share a batch
    // and which have finished when the run stops, so run-to-run digests are only reproducible at 0.
    int SolverThreads = 0;
    uint64_t DfpnNodes = 100000; // node budget of the df-pn mate solver run after every move (worker.cc:516); 0 = off
    bool MateSearch = true;      // mate-in-3 search by checks at every non-root leaf (worker.cc:349-358)
};

struct Stats {
    uint64_t Evaluations = 0;   // leaves sent to the executor
    uint64_t CacheHits = 0;
    uint64_t Batches = 0;
    uint64_t Playouts = 0;      // back-propagations (incl. terminal and cached leaves)
    uint64_t Moves = 0;
    uint64_t MatesFound = 0;    // leaves closed by the mate-in-3 search
    uint64_t DfpnMates = 0;     // games ended by the df-pn solver in judge
    uint64_t DfpnNodes = 0;     // node expansions spent in it
    uint64_t GamesBlack = 0, GamesWhite = 0, GamesDraw = 0;
    uint64_t MovesOfFinishedGames = 0;
    uint64_t TeacherRecords = 0; // positions written by the teacher writer (full-search plies only)
    uint64_t AwaitNs = 0;        // engine thread blocked in Infer::await (the executor still computing)
    uint64_t AwaitsIdle = 0;     // ... of which the batch had ALREADY finished: the executor sat idle
    uint64_t HostNs = 0;         // engine thread applying results and advancing games between two batches
    uint64_t finished() const { return GamesBlack + GamesWhite + GamesDraw; }
};

struct Node;
struct Edge {
    uint16_t Move16;
    float Prior;
    Node* Child;
};
struct Node {
    Node* Parent = nullptr;
    Edge* Edges = nullptr;
    uint32_t Visits = 0;
    uint16_t NumChildren = 0;
    uint8_t Repetition = 0;
    double WinSum = 0.0, DrawSum = 0.0;
    float WinPred = 0.f, DrawPred = 0.f;
};

class Game; // one Frame
class TeacherWriter; // teacher.h
class GameLog;       // teacher.h: one line per finished game
class LeafLog;       // teacher.h: test hook, one line per packed leaf

// One engine = one search thread's worth of games + its two executors.
class Engine {
 public:
    // Exec[0], Exec[1]: one executor per group (not owned).  BatchMax >= GamesPerGroup.
    // Cache: the GPU shard's evaluation cache (shared with the shard's other engines, not owned;
    // nullptr = no cache).  Engine e owns the game slots [e * 2 * GamesPerGroup, (e+1) * 2 * GamesPerGroup).
    Engine(infer::Infer* Exec0, infer::Infer* Exec1, const Options& Opt, uint64_t EngineIndex, bool PinMemory,
           EvalCache* Cache);
    ~Engine();

    // Runs until `Stop` becomes true or `MaxFinishedGames` games have finished (0 = no limit).
    void run(const std::atomic<bool>* Stop, uint64_t MaxFinishedGames);
    // Single-step variant for tests: advance both groups once.
    void step();
    void drain(); // await in-flight batches and apply them

    // Finished games are written to W (shared, not owned; nullptr = off).  Set before run().
    void setTeacherWriter(TeacherWriter* W) { Teacher = W; }
    void setGameLog(GameLog* L) { Log = L; }
    void setLeafLog(LeafLog* L) { Leaves = L; }

    // Progress counters another thread may read while run() is going (the driver's rate timeline)
    uint64_t publishedFinished() const { return PubFinished.load(std::memory_order_relaxed); }
    uint64_t publishedEvaluations() const { return PubEvaluations.load(std::memory_order_relaxed); }
    uint64_t publishedMoves() const { return PubMoves.load(std::memory_order_relaxed); }

    Stats stats() const; // summed over the engine's workers
    // order-independent digest of every move played so far (reproducibility checks): a wrapping sum
    // over (game id, ply, move), so engines' digests add up to the same total however the game slots
    // are spread over them
    uint64_t moveDigest() const;

 private:
    struct Group;
    // What one host thread of the engine owns: its games' counters, mate solver and cache scratch.
    struct WorkerCtx {
        Stats St;
        uint64_t Digest = 0;
        shogi::DfpnSolver Solver;
        EvalCache::Info Scratch;
    };
    void collect(Group& G);
    void apply(Group& G);
    void parallelFor(const std::function<void(int)>& Fn); // Fn(worker) on every worker, caller = worker 0
    void workerLoop(int W);
    void solverLoop();
    void submitSolve(Game* G);
    int ownerOf(std::size_t GameIndex, std::size_t Games) const;

    Options Opt;
    Stats St; // engine-level counters: Batches, Evaluations
    TeacherWriter* Teacher = nullptr;
    GameLog* Log = nullptr;
    LeafLog* Leaves = nullptr;
    uint64_t EngineIndexForLogs = 0;
    std::vector<std::unique_ptr<WorkerCtx>> Ctx;
    std::vector<std::thread> Pool;
    const std::function<void(int)>* Task = nullptr;
    std::atomic<uint64_t> Epoch{0};
    std::atomic<int> Done{0};
    std::atomic<bool> Quit{false};
    std::vector<std::thread> Solvers;
    std::mutex SolveMutex;
    std::condition_variable SolveCV;
    std::deque<Game*> SolveQueue;
    bool SolveQuit = false;
    std::atomic<uint64_t> PubFinished{0}, PubEvaluations{0}, PubMoves{0};
    std::unique_ptr<Group> Groups[2];
    EvalCache* Cache = nullptr;
    friend class Game;
};

} // namespace selfplay
} // namespace engine
} // namespace nshogi

#endif
