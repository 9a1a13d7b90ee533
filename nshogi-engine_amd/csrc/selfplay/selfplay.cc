// selfplay.cc -- see selfplay.h.
#include "selfplay.h"
#include "teacher.h"

#include <nshogi_engine_amd/evaluate/evaluator.h>

#include <algorithm>
#include <cmath>
#include <chrono>
#include <cstring>
#include <limits>
#include <new>

namespace nshogi {
namespace engine {
namespace selfplay {

using shogi::Color;
using shogi::FeaturePlane;
using shogi::Move;
using shogi::MoveList;

namespace {

inline uint64_t mix64(uint64_t X) {
    X ^= X >> 33; X *= 0xff51afd7ed558ccdULL; X ^= X >> 33; X *= 0xc4ceb9fe1a85ec53ULL; X ^= X >> 33;
    return X;
}

// Bump arena holding one game's search tree.  The tree is dropped whole whenever the root
// changes (Tree::updateRoot(State, false), worker.cc:161), so "free" is a rewind: no per-node
// free list, no recursive walk (role of allocator::FixedAllocator<sizeof(Node)> +
// SegregatedFreeListAllocator + the GarbageCollector threads, selfplay/main.cc:72-92).
class Arena {
 public:
    void* alloc(std::size_t Bytes) {
        Bytes = (Bytes + 15) & ~(std::size_t)15;
        if (Bytes > kChunk) throw std::bad_alloc(); // a node's edges: at most 593 * 16 bytes
        if (Chunks.empty() || Off + Bytes > kChunk) {
            if (Cur + 1 < Chunks.size() && !Chunks.empty()) ++Cur;
            else {
                Chunks.emplace_back(new unsigned char[kChunk]);
                Cur = Chunks.size() - 1;
            }
            Off = 0;
        }
        void* P = Chunks[Cur].get() + Off;
        Off += Bytes;
        return P;
    }
    void rewind() {
        Cur = 0;
        Off = 0;
    }

 private:
    static constexpr std::size_t kChunk = 256 * 1024;
    std::vector<std::unique_ptr<unsigned char[]>> Chunks;
    std::size_t Cur = 0, Off = 0;
};

} // namespace

class Game {
 public:
    // Slot: this game slot's number among the run's Stride concurrent slots; its k-th game has id Slot + k * Stride
    Game(Engine* E, Engine::WorkerCtx* C, uint64_t Slot, uint64_t Stride) : Eng(E), Ctx(C), GameId(Slot), Stride(Stride) { newGame(false); }
    // the host thread about to advance this game lends it its counters, solver and cache scratch
    void bind(Engine::WorkerCtx* C) { Ctx = C; }
    Game(const Game&) = delete;
    Game& operator=(const Game&) = delete;

    // Advances the state machine until a leaf needs the network; writes its feature
    // planes to `Slot` and returns true.  Returns false if the budget of host-only
    // playouts for this call is used up (everything was terminal or cached).
    bool advanceUntilEvaluation(FeaturePlane* Slot) {
        for (int Guard = 0; Guard < 20000; ++Guard) {
            switch (Ph) {
            case Phase::RootPreparation: prepareRoot(); break;
            case Phase::LeafSelection: selectLeaf(); break;
            case Phase::LeafTerminalChecking:
                if (checkTerminal()) {
                    shogi::buildFeatures(S, Config, Slot);
                    Ph = Phase::Evaluation;
                    return true;
                }
                break;
            case Phase::Evaluation: return false; // still waiting (should not happen)
            case Phase::Backpropagation: backpropagate(); break;
            case Phase::Transition: transition(); break;
            case Phase::Judging: judge(); break;
            case Phase::SolverWait:
                if (!SolverDone.load(std::memory_order_acquire)) return false; // the mate solver still has this position
                afterSolver(SolverMove, SolverNodes);
                break;
            }
        }
        return false;
    }

    // (leaf log) the position and StateConfig of the leaf waiting for its evaluation
    std::string leafSfen() const { return S.toSfen(); }
    const shogi::StateConfig& config() const { return Config; }

    // Frame::setEvaluation<false> (frame.cc:93-136) for the pending leaf.
    void setEvaluation(const float* Policy, float Win, float Draw) {
        const uint16_t N = Leaf->NumChildren;
        const Color Us = S.sideToMove();
        for (uint16_t I = 0; I < N; ++I) {
            Move M;
            M.V = Leaf->Edges[I].Move16;
            Logits[I] = Policy[shogi::moveIndex(Us, M)];
        }
        if (Eng->Cache) Eng->Cache->store(S.hash(), N, Logits, Win, Draw); // frame.cc:108-111
        finishEvaluation(N, Win, Draw);
    }

 private:
    enum class Phase { RootPreparation, LeafSelection, LeafTerminalChecking, Evaluation, Backpropagation, Transition, Judging, SolverWait };

 public:
    // solver-pool side: the game is parked in SolverWait, nobody else touches S meanwhile
    void solveNow(shogi::DfpnSolver& Solver) {
        SolverMove = Solver.solve(S, Eng->Opt.DfpnNodes);
        SolverNodes = Solver.nodes();
        SolverDone.store(true, std::memory_order_release);
    }

 private:

    Node* newNode(Node* Parent) {
        Node* N = new (Tree.alloc(sizeof(Node))) Node();
        N->Parent = Parent;
        return N;
    }

    void newGame(bool Next = true) {
        Tree.rewind();
        Root = nullptr;
        S = shogi::State();
        if (Next) GameId += Stride;
        Rng.seed(mix64(Eng->Opt.Seed ^ mix64(GameId * 0x9e3779b97f4a7c15ULL + 1)));
        // worker.cc:132-150
        std::uniform_int_distribution<int> MaxPly(Eng->Opt.MaxPlyMin, Eng->Opt.MaxPlyMax);
        Config.MaxPly = (uint16_t)MaxPly(Rng);
        Config.BlackDrawValue = Config.WhiteDrawValue = 0.5f;
        if (Eng->Opt.RandomDrawValue && (Rng() % 4) >= 2) {
            Config.BlackDrawValue = std::uniform_real_distribution<float>(0.f, 1.f)(Rng);
            Config.WhiteDrawValue = 1.0f - Config.BlackDrawValue;
        }
        GameMoves = 0;
        MoveHistory.clear();
        FullSearchAt.clear();
        Ph = Phase::RootPreparation;
    }

    float drawValue(Color C) const { return C == shogi::Black ? Config.BlackDrawValue : Config.WhiteDrawValue; }

    void prepareRoot() { // worker.cc:159-215
        Tree.rewind();
        Root = newNode(nullptr);
        RootPly = S.ply();
        MoveList L;
        S.generateLegalMoves(L);
        const bool Gumbel = Eng->Opt.Gumbel;
        if (Gumbel) { // worker.cc:163-165, 640-648: one Gumbel(0,1) sample per root move
            std::uniform_real_distribution<double> U(std::numeric_limits<double>::min(), 1.0);
            for (int I = 0; I < 600; ++I) Noise[I] = -std::log(-std::log(U(Rng)));
        }
        const double R = std::uniform_real_distribution<double>(0.0, 1.0)(Rng);
        if (L.size() == 1 || R > Eng->Opt.FullSearchRatio) {
            if (Gumbel) Budget = (uint32_t)Eng->Opt.NumSamplingMoves;
            else Budget = L.size() == 1 ? 1u : (uint32_t)std::max(1, Eng->Opt.NumPlayouts / 4);
            FullSearch = false;
        } else {
            Budget = (uint32_t)Eng->Opt.NumPlayouts;
            FullSearch = true;
        }
        if (Gumbel) { // worker.cc:203-212
            NumSampling = (uint32_t)Eng->Opt.NumSamplingMoves;
            const double Init = std::floor((double)Budget / (std::log2((double)NumSampling) * (double)NumSampling));
            HalvingPlayouts = (uint32_t)std::max(1.0, Init);
            HalvingCount = 1;
        }
        Ph = Phase::LeafSelection;
    }

    void undoToRoot() {
        while (S.ply() > RootPly) S.undoMove();
    }

    // worker.cc:726-770
    double winRateOfChild(Color SideToMove, const Node* Child) const {
        const double V = (double)Child->Visits;
        const double WinRate = (V - Child->WinSum) / V;
        const double DrawRate = Child->DrawSum / V;
        return DrawRate * (double)drawValue(SideToMove) + (1.0 - DrawRate) * WinRate;
    }

    Edge* pickEdge(Node* N) const { // PUCT, worker.cc:688-715
        const double C = 1.25 * std::sqrt((double)N->Visits);
        const Color Us = S.sideToMove();
        double Best = std::numeric_limits<double>::lowest();
        Edge* BestEdge = nullptr;
        for (uint16_t I = 0; I < N->NumChildren; ++I) {
            Edge* E = &N->Edges[I];
            const Node* Ch = E->Child;
            const double Score = (Ch == nullptr || Ch->Visits == 0)
                                     ? C * (double)E->Prior
                                     : C * (double)E->Prior / (double)(1 + Ch->Visits) + winRateOfChild(Us, Ch);
            if (Score > Best) {
                Best = Score;
                BestEdge = E;
            }
        }
        return BestEdge;
    }

    Edge* pickGumbelRootEdge() const { // worker.cc:668-686: next target below the halving quota
        for (uint16_t I = 0; I < Root->NumChildren; ++I) {
            if (!IsTarget[I]) continue;
            Edge* E = &Root->Edges[I];
            if (E->Child == nullptr || E->Child->Visits < HalvingPlayouts) return E;
        }
        // every target already has its quota (can happen once the budget is spent): any target
        for (uint16_t I = 0; I < Root->NumChildren; ++I)
            if (IsTarget[I]) return &Root->Edges[I];
        return &Root->Edges[0];
    }

    static double transformQ(double Q, uint32_t MaxN) { return (50.0 + (double)MaxN) * 1.0 * Q; } // worker.cc:654-659

    double gumbelScore(uint16_t I) const {
        const Edge* E = &Root->Edges[I];
        return Noise[I] + (double)E->Prior + transformQ(winRateOfChild(S.sideToMove(), E->Child), E->Child->Visits);
    }

    void sampleTopMMoves() { // worker.cc:784-817
        const uint16_t N = Root->NumChildren;
        IsTarget.assign(N, true);
        if (NumSampling >= N) return;
        std::vector<std::pair<double, uint16_t>> Score(N);
        for (uint16_t I = 0; I < N; ++I) Score[I] = {Noise[I] + (double)Root->Edges[I].Prior, I};
        std::partial_sort(Score.begin(), Score.begin() + NumSampling, Score.end(),
                          [](const auto& A, const auto& B) { return A.first > B.first; });
        IsTarget.assign(N, false);
        for (uint32_t I = 0; I < NumSampling; ++I) IsTarget[Score[I].second] = true;
    }

    uint16_t executeSequentialHalving() { // worker.cc:819-863
        const uint16_t N = Root->NumChildren;
        std::vector<std::pair<double, uint16_t>> Score(N);
        for (uint16_t I = 0; I < N; ++I)
            Score[I] = {IsTarget[I] ? gumbelScore(I) : std::numeric_limits<double>::lowest(), I};
        std::size_t NumSort = std::min<std::size_t>(NumSampling, N);
        NumSort = std::max<std::size_t>(2, (NumSort + 1) >> HalvingCount);
        std::partial_sort(Score.begin(), Score.begin() + (long)NumSort, Score.end(),
                          [](const auto& A, const auto& B) { return A.first > B.first; });
        IsTarget.assign(N, false);
        for (std::size_t I = 0; I < NumSort; ++I) IsTarget[Score[I].second] = true;
        return (uint16_t)NumSort;
    }

    bool updateHalvingSchedule(uint16_t NumValid) { // worker.cc:865-905
        const uint32_t MD = std::max<uint32_t>(1, NumSampling >> HalvingCount);
        const double D = std::log2((double)NumSampling) * (double)MD;
        const uint64_t Extra = (uint64_t)std::floor((double)Budget / D);
        if (Extra == 0) return false;
        if (NumValid <= 2) {
            const uint64_t Left = (uint64_t)Budget + 1 - Root->Visits;
            HalvingPlayouts += (uint32_t)((Left + 1) / 2);
        } else {
            HalvingPlayouts += (uint32_t)Extra;
        }
        ++HalvingCount;
        return true;
    }

    // the Gumbel branch of worker.cc:412-475
    void gumbelAfterBackprop() {
        if (Root->NumChildren == 1) { Ph = Phase::Transition; return; }
        if (Leaf == Root) {
            sampleTopMMoves();
        } else {
            uint32_t MinN = std::numeric_limits<uint32_t>::max();
            for (uint16_t I = 0; I < Root->NumChildren; ++I) {
                if (!IsTarget[I]) continue;
                const Node* Ch = Root->Edges[I].Child;
                if (Ch == nullptr) { MinN = 0; break; }
                MinN = std::min(MinN, Ch->Visits);
            }
            if (MinN >= HalvingPlayouts) {
                if (Root->Visits >= Budget + 1) { Ph = Phase::Transition; return; }
                const uint16_t NumValid = executeSequentialHalving();
                if (!updateHalvingSchedule(NumValid)) { Ph = Phase::Transition; return; }
            }
        }
        Ph = Phase::LeafSelection;
    }

    void selectLeaf() { // worker.cc:217-266
        undoToRoot();
        Node* N = Root;
        for (;;) {
            if (N->Visits == 0 || N->NumChildren == 0 || N->Repetition != 0) break;
            if (S.ply() >= Config.MaxPly) break;
            Edge* E = (Eng->Opt.Gumbel && N == Root) ? pickGumbelRootEdge() : pickEdge(N);
            S.doMove(S.moveFrom16(E->Move16));
            if (!E->Child) E->Child = newNode(N);
            N = E->Child;
        }
        Leaf = N;
        Ph = Phase::LeafTerminalChecking;
    }

    void setTerminal(float Win, float Draw) { // setEvaluation<true>(nullptr, ...)
        Leaf->WinPred = Win;
        Leaf->DrawPred = Draw;
        Ph = Phase::Backpropagation;
    }

    // returns true when the leaf needs a network evaluation; worker.cc:268-381
    bool checkTerminal() {
        if (Leaf->Visits > 0) {
            Ph = Phase::Backpropagation;
            return false;
        }
        const shogi::RepetitionStatus RS = S.repetitionStatus(true);
        if (RS != shogi::NoRepetition) {
            Leaf->Repetition = (uint8_t)RS;
            if (RS == shogi::WinRepetition) setTerminal(1.0f, 0.0f);
            else if (RS == shogi::LossRepetition) setTerminal(0.0f, 0.0f);
            else setTerminal(drawValue(S.sideToMove()), 1.0f);
            return false;
        }
        if (Config.Declare27 && Leaf != Root && S.canDeclare()) {
            setTerminal(1.0f, 0.0f);
            return false;
        }
        MoveList L;
        S.generateLegalMoves(L);
        if (L.size() == 0) {
            setTerminal(0.0f, 0.0f);
            return false;
        }
        if (S.ply() >= Config.MaxPly) {
            setTerminal(drawValue(S.sideToMove()), 1.0f);
            return false;
        }
        // checkmate by search (worker.cc:349-358: solver::dfs::solve(State, 3) at every non-root leaf)
        if (Eng->Opt.MateSearch && Leaf != Root && !S.findMate(3, true, &L).isNone()) {
            ++Ctx->St.MatesFound;
            setTerminal(1.0f, 0.0f);
            return false;
        }
        // expand (Node::expand)
        Leaf->NumChildren = (uint16_t)L.size();
        Leaf->Edges = static_cast<Edge*>(Tree.alloc(sizeof(Edge) * (std::size_t)L.size()));
        for (int I = 0; I < L.size(); ++I) Leaf->Edges[I] = Edge{L[I].move16(), 0.0f, nullptr};
        if (Eng->Cache) { // worker.cc:367-378
            EvalCache::Info& E = Ctx->Scratch;
            if (Eng->Cache->load(S.hash(), &E) && E.NumMoves == Leaf->NumChildren) {
                std::memcpy(Logits, E.Policy, Leaf->NumChildren * sizeof(float));
                ++Ctx->St.CacheHits;
                finishEvaluation(Leaf->NumChildren, E.WinRate, E.DrawRate);
                return false;
            }
        }
        return true;
    }

    // softmax over the legal-move logits, Dirichlet noise at a full-search root
    // (frame.cc:113-135), priors into the edges
    void finishEvaluation(uint16_t N, float Win, float Draw) {
        if (Eng->Opt.Gumbel && Leaf == Root) { // frame.cc:116: a Gumbel root keeps the raw logits
            for (uint16_t I = 0; I < N; ++I) Leaf->Edges[I].Prior = Logits[I];
            Leaf->WinPred = Win;
            Leaf->DrawPred = Draw;
            Ph = Phase::Backpropagation;
            return;
        }
        float Max = -std::numeric_limits<float>::infinity();
        for (uint16_t I = 0; I < N; ++I) Max = std::max(Max, Logits[I]);
        float Sum = 0.f;
        for (uint16_t I = 0; I < N; ++I) {
            Logits[I] = std::exp(Logits[I] - Max);
            Sum += Logits[I];
        }
        for (uint16_t I = 0; I < N; ++I) Logits[I] /= Sum;
        if (Leaf == Root && FullSearch) {
            std::gamma_distribution<double> Gamma(0.15, 1.0); // worker.cc:650-652
            double NoiseSum = 0.0;
            for (uint16_t I = 0; I < N; ++I) {
                Noise[I] = Gamma(Rng);
                NoiseSum += Noise[I];
            }
            if (NoiseSum <= 0.0) NoiseSum = 1.0;
            constexpr double Eps = 0.25;
            for (uint16_t I = 0; I < N; ++I)
                Logits[I] = (float)((1 - Eps) * (double)Logits[I] + Eps * Noise[I] / NoiseSum);
        }
        for (uint16_t I = 0; I < N; ++I) Leaf->Edges[I].Prior = Logits[I];
        Leaf->WinPred = Win;
        Leaf->DrawPred = Draw;
        Ph = Phase::Backpropagation;
    }

    void backpropagate() { // node.h:170-202, worker.cc:383-426
        const float Win = Leaf->WinPred, Draw = Leaf->DrawPred;
        bool Flip = false;
        for (Node* N = Leaf; N != nullptr; N = N->Parent, Flip = !Flip) {
            N->WinSum += Flip ? 1.0 - (double)Win : (double)Win;
            N->DrawSum += (double)Draw;
            ++N->Visits;
        }
        undoToRoot();
        ++Ctx->St.Playouts;
        if (Eng->Opt.Gumbel) gumbelAfterBackprop();
        else if (Root->NumChildren == 1 || Root->Visits >= Budget) Ph = Phase::Transition;
        else Ph = Phase::LeafSelection;
    }

    void transition() { // most visited move, ties by prior (worker.cc:562-596)
        undoToRoot();
        if (Eng->Opt.Gumbel) { // worker.cc:598-638
            Edge* Pick = &Root->Edges[0];
            if (Root->NumChildren > 1) {
                double BestScore = std::numeric_limits<double>::lowest();
                for (uint16_t I = 0; I < Root->NumChildren; ++I) {
                    if (!IsTarget[I] || Root->Edges[I].Child == nullptr || Root->Edges[I].Child->Visits == 0) continue;
                    const double Sc = gumbelScore(I);
                    if (Sc > BestScore) { BestScore = Sc; Pick = &Root->Edges[I]; }
                }
            }
            playMove(S.moveFrom16(Pick->Move16));
            return;
        }
        uint32_t MaxVisits = 0;
        Edge* Best = nullptr;
        for (uint16_t I = 0; I < Root->NumChildren; ++I) {
            Edge* E = &Root->Edges[I];
            const Node* Ch = E->Child;
            if (Ch == nullptr || Ch->Visits == 0) {
                if (Best == nullptr) Best = E;
                else if ((Best->Child == nullptr || Best->Child->Visits == 0) && E->Prior > Best->Prior) Best = E;
                continue;
            }
            if (Ch->Visits > MaxVisits) {
                MaxVisits = Ch->Visits;
                Best = E;
            } else if (Ch->Visits == MaxVisits && E->Prior > Best->Prior) {
                Best = E;
            }
        }
        playMove(S.moveFrom16(Best->Move16));
    }

    void playMove(Move M) {
        MoveHistory.push_back(M.V);
        FullSearchAt.push_back(FullSearch ? 1 : 0); // Frame::DidFullSearch (frame.h), read by saveworker.cc:172-174
        S.doMove(M);
        ++GameMoves;
        ++Ctx->St.Moves;
        Ctx->Digest += mix64(mix64(GameId + 0x51ed270b) ^ ((uint64_t)GameMoves << 32) ^ M.V);
        Ph = Phase::Judging;
    }

    void finish(Color Winner) {
        if (Winner == shogi::Black) ++Ctx->St.GamesBlack;
        else if (Winner == shogi::White) ++Ctx->St.GamesWhite;
        else ++Ctx->St.GamesDraw;
        Ctx->St.MovesOfFinishedGames += GameMoves;
        if (Eng->Teacher) Ctx->St.TeacherRecords += Eng->Teacher->saveGame(MoveHistory, FullSearchAt, Config, Winner);
        if (Eng->Log) Eng->Log->add(GameId, Winner, MoveHistory);
        newGame();
    }

    void judge() { // worker.cc:477-526
        const shogi::RepetitionStatus RS = S.repetitionStatus(true);
        if (RS == shogi::WinRepetition) return finish(S.sideToMove());
        if (RS == shogi::LossRepetition) return finish(~S.sideToMove());
        if (RS == shogi::Repetition) return finish(shogi::NoColor);
        if (Config.Declare27 && S.canDeclare()) return finish(S.sideToMove());
        MoveList L;
        S.generateLegalMoves(L);
        if (L.size() == 0) return finish(~S.sideToMove());
        if (S.ply() >= Config.MaxPly) return finish(shogi::NoColor);
        if (Eng->Opt.DfpnNodes) { // worker.cc:516-524: a proven mate ends the game with the mating move played
            if (Eng->Opt.SolverThreads > 0) {
                SolverDone.store(false, std::memory_order_relaxed);
                Ph = Phase::SolverWait;
                Eng->submitSolve(this);
                return;
            }
            const Move Mate = Ctx->Solver.solve(S, Eng->Opt.DfpnNodes);
            return afterSolver(Mate, Ctx->Solver.nodes());
        }
        Ph = Phase::RootPreparation;
    }

    void afterSolver(Move Mate, uint64_t Nodes) {
        Ctx->St.DfpnNodes += Nodes;
        if (!Mate.isNone()) {
            const Color Winner = S.sideToMove();
            FullSearch = true; // pushDidFullSearch(true)
            playMove(Mate);
            ++Ctx->St.DfpnMates;
            return finish(Winner);
        }
        Ph = Phase::RootPreparation;
    }

    std::atomic<bool> SolverDone{false};
    Move SolverMove;
    uint64_t SolverNodes = 0;
    Engine* Eng;
    Engine::WorkerCtx* Ctx; // the worker advancing this game right now (bind)
    uint64_t GameId; // of the game being played: slot + k * Stride
    uint64_t Stride;
    Arena Tree;
    shogi::State S;
    shogi::StateConfig Config;
    std::mt19937_64 Rng;
    Node* Root = nullptr;
    Node* Leaf = nullptr;
    int RootPly = 0;
    uint32_t Budget = 1;
    bool FullSearch = false;
    uint32_t GameMoves = 0;
    std::vector<uint32_t> MoveHistory;  // every move of the current game (for the teacher replay)
    std::vector<uint8_t> FullSearchAt;  // per ply: was the root search a full search
    Phase Ph = Phase::RootPreparation;
    float Logits[600];
    double Noise[600];
    // Gumbel state (frame.h: IsTarget, NumSamplingMoves, SequentialHalvingPlayouts/Count)
    std::vector<bool> IsTarget;
    uint32_t NumSampling = 16, HalvingPlayouts = 1, HalvingCount = 1;
};

struct Engine::Group {
    std::vector<std::unique_ptr<Game>> Games;
    std::unique_ptr<evaluate::Evaluator> Ev;
    std::vector<int> Pending;
    std::vector<uint8_t> Has;        // collect: game I produced a leaf
    std::unique_ptr<std::atomic<std::size_t>[]> Cursor; // per worker range: next item to take (work stealing)
    std::size_t Count = 0;
    bool InFlight = false;
};

int Engine::ownerOf(std::size_t GameIndex, std::size_t Games) const {
    // worker W owns the games [Games * W / Workers, Games * (W + 1) / Workers)
    const std::size_t W = (std::size_t)Opt.Workers;
    std::size_t Guess = GameIndex * W / Games;
    while (Guess + 1 < W && Games * (Guess + 1) / W <= GameIndex) ++Guess;
    while (Guess > 0 && Games * Guess / W > GameIndex) --Guess;
    return (int)Guess;
}

Engine::Engine(infer::Infer* Exec0, infer::Infer* Exec1, const Options& O, uint64_t EngineIndex, bool PinMemory,
               EvalCache* SharedCache)
    : Opt(O), Cache(SharedCache) {
    EngineIndexForLogs = EngineIndex;
    if (Opt.Workers < 1) Opt.Workers = 1;
    if (Opt.Workers > Opt.GamesPerGroup) Opt.Workers = Opt.GamesPerGroup;
    for (int W = 0; W < Opt.Workers; ++W) Ctx.push_back(std::make_unique<WorkerCtx>());
    infer::Infer* Exec[2] = {Exec0, Exec1};
    const uint64_t Mine = 2 * (uint64_t)Opt.GamesPerGroup;
    const uint64_t Stride = Opt.TotalSlots ? Opt.TotalSlots : Mine;
    for (int G = 0; G < 2; ++G) {
        Groups[G] = std::make_unique<Group>();
        Groups[G]->Ev = std::make_unique<evaluate::Evaluator>(EngineIndex * 2 + G, shogi::NumFeaturePlanes,
                                                              (std::size_t)Opt.GamesPerGroup, Exec[G], PinMemory);
        Groups[G]->Pending.resize(Opt.GamesPerGroup);
        Groups[G]->Has.assign((std::size_t)Opt.GamesPerGroup, 0);
        Groups[G]->Cursor = std::make_unique<std::atomic<std::size_t>[]>((std::size_t)Opt.Workers);
        for (int I = 0; I < Opt.GamesPerGroup; ++I)
            Groups[G]->Games.push_back(std::make_unique<Game>(
                this, Ctx[(std::size_t)ownerOf((std::size_t)I, (std::size_t)Opt.GamesPerGroup)].get(),
                EngineIndex * Mine + (uint64_t)G * (uint64_t)Opt.GamesPerGroup + (uint64_t)I, Stride));
    }
    for (int W = 1; W < Opt.Workers; ++W) Pool.emplace_back([this, W]() { workerLoop(W); });
    for (int T = 0; T < Opt.SolverThreads; ++T) Solvers.emplace_back([this]() { solverLoop(); });
}

void Engine::submitSolve(Game* G) {
    {
        std::lock_guard<std::mutex> Lock(SolveMutex);
        SolveQueue.push_back(G);
    }
    SolveCV.notify_one();
}

void Engine::solverLoop() {
    shogi::DfpnSolver Solver;
    for (;;) {
        Game* G;
        {
            std::unique_lock<std::mutex> Lock(SolveMutex);
            SolveCV.wait(Lock, [this]() { return SolveQuit || !SolveQueue.empty(); });
            if (SolveQueue.empty()) return;
            G = SolveQueue.front();
            SolveQueue.pop_front();
        }
        G->solveNow(Solver);
    }
}

Engine::~Engine() {
    drain();
    {
        std::lock_guard<std::mutex> Lock(SolveMutex);
        SolveQuit = true;
        SolveQueue.clear(); // games parked on the solver are simply never resumed
    }
    SolveCV.notify_all();
    for (auto& T : Solvers) T.join();
    Quit.store(true, std::memory_order_release);
    Epoch.fetch_add(1, std::memory_order_release);
    for (auto& T : Pool) T.join();
}

void Engine::workerLoop(int W) {
    uint64_t Seen = 0;
    for (;;) {
        // steps are about a millisecond apart: spin, but give the core away now and then
        for (unsigned Spin = 0; Epoch.load(std::memory_order_acquire) == Seen; ++Spin)
            if ((Spin & 63) == 63) std::this_thread::yield();
        ++Seen;
        if (Quit.load(std::memory_order_acquire)) return;
        (*Task)(W);
        Done.fetch_add(1, std::memory_order_release);
    }
}

void Engine::parallelFor(const std::function<void(int)>& Fn) {
    if (Pool.empty()) {
        Fn(0);
        return;
    }
    Task = &Fn;
    Done.store(0, std::memory_order_relaxed);
    Epoch.fetch_add(1, std::memory_order_release);
    Fn(0);
    while (Done.load(std::memory_order_acquire) < (int)Pool.size()) {
    }
}

Stats Engine::stats() const {
    Stats S = St;
    for (const auto& C : Ctx) {
        const Stats& X = C->St;
        S.CacheHits += X.CacheHits; S.Playouts += X.Playouts; S.Moves += X.Moves; S.MatesFound += X.MatesFound;
        S.DfpnMates += X.DfpnMates; S.DfpnNodes += X.DfpnNodes; S.GamesBlack += X.GamesBlack;
        S.GamesWhite += X.GamesWhite; S.GamesDraw += X.GamesDraw; S.MovesOfFinishedGames += X.MovesOfFinishedGames;
        S.TeacherRecords += X.TeacherRecords;
    }
    return S;
}

uint64_t Engine::moveDigest() const {
    uint64_t D = 0;
    for (const auto& C : Ctx) D += C->Digest;
    return D;
}

void Engine::apply(Group& G) {
    if (!G.InFlight) return;
    const auto T0 = std::chrono::steady_clock::now();
    if (!G.Ev->isComputing()) ++St.AwaitsIdle;
    G.Ev->await();
    St.AwaitNs += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - T0).count();
    // (the pending list is in game order: worker w starts on its share of it, the same games it will
    // advance first in the next collect, and helps with the other shares when it is done)
    const std::size_t W = (std::size_t)Opt.Workers, Cnt = G.Count;
    for (std::size_t Wk = 0; Wk < W; ++Wk) G.Cursor[Wk].store(Cnt * Wk / W, std::memory_order_relaxed);
    parallelFor([&](int Me) {
        WorkerCtx* Mine = Ctx[(std::size_t)Me].get();
        for (std::size_t Step = 0; Step < W; ++Step) {
            const std::size_t Wk = ((std::size_t)Me + Step) % W;
            const std::size_t Hi = Cnt * (Wk + 1) / W;
            for (;;) {
                const std::size_t K = G.Cursor[Wk].fetch_add(1, std::memory_order_relaxed);
                if (K >= Hi) break;
                Game& Gm = *G.Games[(std::size_t)G.Pending[K]];
                Gm.bind(Mine);
                Gm.setEvaluation(G.Ev->getPolicy() + K * shogi::MoveIndexMax, G.Ev->getWinRate()[K], G.Ev->getDrawRate()[K]);
            }
        }
    });
    G.InFlight = false;
}

void Engine::collect(Group& G) {
    static_assert(sizeof(FeaturePlane) == sizeof(ml::FeatureBitboard), "feature plane layout");
    auto* Slots = reinterpret_cast<FeaturePlane*>(G.Ev->getFeatureBitboards());
    const std::size_t Games = G.Games.size(), W = (std::size_t)Opt.Workers;
    // Worker w advances the games of its own range first (their trees are warm in its caches), then
    // takes games from the other ranges until none is left: a leaf costs anything between a cache hit
    // and a three-ply mate search, and with fixed ranges every batch waited for the unluckiest worker.
    // Which thread advances a game changes nothing in it (counters and digests are sums).  A game's
    // planes go to the slot of its own index; the gaps (a game parked on the mate solver) are closed
    // afterwards, so the batch order is the game order whatever the threads did.
    for (std::size_t Wk = 0; Wk < W; ++Wk) G.Cursor[Wk].store(Games * Wk / W, std::memory_order_relaxed);
    parallelFor([&](int Me) {
        WorkerCtx* Mine = Ctx[(std::size_t)Me].get();
        for (std::size_t Step = 0; Step < W; ++Step) {
            const std::size_t Wk = ((std::size_t)Me + Step) % W;
            const std::size_t Hi = Games * (Wk + 1) / W;
            for (;;) {
                const std::size_t I = G.Cursor[Wk].fetch_add(1, std::memory_order_relaxed);
                if (I >= Hi) break;
                Game& Gm = *G.Games[I];
                Gm.bind(Mine);
                G.Has[I] = Gm.advanceUntilEvaluation(Slots + I * shogi::NumFeaturePlanes) ? 1 : 0;
            }
        }
    });
    std::size_t N = 0;
    for (std::size_t I = 0; I < Games; ++I) {
        if (!G.Has[I]) continue;
        if (I != N)
            std::memcpy(Slots + N * shogi::NumFeaturePlanes, Slots + I * shogi::NumFeaturePlanes,
                        shogi::NumFeaturePlanes * sizeof(FeaturePlane));
        G.Pending[N++] = (int)I;
    }
    G.Count = N;
    if (N == 0) return;
    if (Leaves) { // test hook: what sits in every slot of the packed batch, and the position it should encode
        for (std::size_t K = 0; K < N; ++K) {
            const Game& Gm = *G.Games[(std::size_t)G.Pending[K]];
            Leaves->add(EngineIndexForLogs * 2 + (std::size_t)(&G == Groups[1].get()), St.Batches, K, Gm.leafSfen(), Gm.config().MaxPly,
                        Gm.config().BlackDrawValue, Slots + K * shogi::NumFeaturePlanes,
                        shogi::NumFeaturePlanes * sizeof(FeaturePlane));
        }
    }
    G.Ev->computeNonBlocking(N);
    G.InFlight = true;
    ++St.Batches;
    St.Evaluations += N;
}

void Engine::step() {
    for (int G = 0; G < 2; ++G) {
        const uint64_t W0 = St.AwaitNs;
        const auto T0 = std::chrono::steady_clock::now();
        apply(*Groups[G]);   // results of this group's previous batch
        collect(*Groups[G]); // host search of this group while the other group's batch computes
        St.HostNs += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - T0).count() -
                     (St.AwaitNs - W0);
    }
    const Stats S = stats();
    PubFinished.store(S.finished(), std::memory_order_relaxed);
    PubEvaluations.store(S.Evaluations, std::memory_order_relaxed);
    PubMoves.store(S.Moves, std::memory_order_relaxed);
}

void Engine::drain() {
    for (int G = 0; G < 2; ++G) apply(*Groups[G]);
}

void Engine::run(const std::atomic<bool>* Stop, uint64_t MaxFinishedGames) {
    while (!Stop->load(std::memory_order_relaxed) && (MaxFinishedGames == 0 || stats().finished() < MaxFinishedGames)) step();
    drain();
}

} // namespace selfplay
} // namespace engine
} // namespace nshogi
