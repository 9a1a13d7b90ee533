// nsg_capi.hip -- implementation of the C ABI declared in include/nsg.h.
//
// One nsg_evaluator = one (device, stream) executor, the role the reference
// gives to infer::TensorRT (src/infer/trt.h:42-88, src/infer/trt.cc).  The
// forward pass it enqueues replaces trt.cc:234-272:
//     H2D bitboards -> feature planes -> policy/value/draw network -> 3x D2H
// all on the evaluator's own non-blocking stream.  No CPU fallback exists:
// every compute entry fails with NSG_E_HIP when no device is usable.
#include "../../include/nsg.h"
#include "kernels/kernels.h"
#include "onnx_reader.h"

#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <fcntl.h>
#include <strings.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string gLastError;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    gLastError = buf;
    return code;
}

#define NSG_HIP(call)                                                             \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess)                                                     \
            return fail(NSG_E_HIP, "%s failed: %s (%s:%d)", #call,                \
                        hipGetErrorString(e_), __FILE__, __LINE__);               \
    } while (0)

// rocprofv3 markers (the reference has no profiler hooks; the build adds them): with
// NSG_ROCTX=1 every forward pass brackets its phases -- H2D, planes, trunk, heads, D2H -- with roctx
// ranges, resolved from librocprofiler-sdk-roctx.so at first use (no link-time dependency; without the
// variable, or without the library, a range is two predictable branches).  The ranges mark where the
// phase is ENQUEUED on the host; `rocprofv3 --marker-trace --kernel-trace` lines them up with the kernels.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("NSG_ROCTX");
        if (!e || e[0] == '0') return;
        void* h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("libroctx64.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) return;
        push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        pop = (int (*)())dlsym(h, "roctxRangePop");
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const Roctx& roctx() {
    static const Roctx r;
    return r;
}
struct Range {
    bool on;
    explicit Range(const char* name) : on(roctx().push != nullptr) {
        if (on) roctx().push(name);
    }
    ~Range() {
        if (on) roctx().pop();
    }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    int alloc(size_t n, bool zero) {
        release();
        if (n == 0) n = 16;
        hipError_t e = hipMalloc(&p, n);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(NSG_E_HIP, "hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        }
        bytes = n;
        if (zero) {
            e = hipMemset(p, 0, n);
            if (e != hipSuccess) return fail(NSG_E_HIP, "hipMemset failed: %s", hipGetErrorString(e));
        }
        return NSG_OK;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    ~DevBuf() { release(); }
};

struct ConvLayer {
    DevBuf w;    // fragment-ordered weights
    DevBuf bias; // f32 [cout]
    int cin = 0; // padded input channels
    float accScale = 1.f; // 1 / (power-of-two weight scale), kF16x3 only
};

// Everything nsg_load* uploads: read-only after the load, so evaluators on one device share
// one copy (nsg_load_shared) and evaluators on other devices take a peer copy of it.
struct NetWeights {
    int gpu = 0;
    int prec = 0;
    int F = 0, blocks = 0, vc = 0, vh = 0, cin = 0, cpad = 0, headsCout = 0, fc1K = 0;
    uint64_t params = 0;
    // Estimated largest trunk activation (estimateActivationBound) and, for kF16m8, whether it leaves the
    // window its fixed-scale e4m3 copies cover: then every batch runs the kF16x3 copy of the trunk.
    double actBound = 0.0;
    bool outsideM8Window = false;
    ConvLayer stem;
    std::vector<ConvLayer> conv1, conv2;
    // kF16m8 also holds the trunk in kF16x3 form: batches too small for full
    // tiles (4 fragments per wave) run the kF16x3 small-tile kernels instead
    ConvLayer stemX3;
    std::vector<ConvLayer> conv1X3, conv2X3;
    ConvLayer heads;
    ConvLayer fc1;
    DevBuf fc2W, fc2B;
    ~NetWeights() {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        (void)hipSetDevice(gpu); // the buffers are released on the device that owns them
        stem = ConvLayer(); stemX3 = ConvLayer(); heads = ConvLayer(); fc1 = ConvLayer();
        conv1.clear(); conv2.clear(); conv1X3.clear(); conv2X3.clear();
        fc2W.release(); fc2B.release();
        if (cur >= 0) (void)hipSetDevice(cur);
    }
};

// Parsed view of an NSGW v1 blob (DESIGN.md "Weight file").
struct NetView {
    int cin, F, blocks, pc, vc, vh;
    float eps;
    const float* stemW; const float* stemBn;
    std::vector<const float*> w1, bn1, w2, bn2;
    const float* polW; const float* polB;
    const float* valW; const float* valBn;
    const float* fc1W; const float* fc1B;
    const float* fc2W; const float* fc2B;
    uint64_t params;
};

int parseBlob(const void* blob, size_t size, NetView* nv) {
    if (size < 64) return fail(NSG_E_FORMAT, "weight blob too small (%zu bytes)", size);
    const unsigned char* p = (const unsigned char*)blob;
    if (memcmp(p, "NSGW", 4) != 0) return fail(NSG_E_FORMAT, "bad magic (not an NSGW file)");
    uint32_t h[15];
    memcpy(h, p + 4, sizeof(h));
    if (h[0] != 1) return fail(NSG_E_FORMAT, "unsupported NSGW version %u", h[0]);
    nv->cin = (int)h[1]; nv->F = (int)h[2]; nv->blocks = (int)h[3];
    nv->pc = (int)h[4]; nv->vc = (int)h[5]; nv->vh = (int)h[6];
    memcpy(&nv->eps, &h[7], 4);
    const size_t F = nv->F, C = nv->cin, VC = nv->vc, VH = nv->vh, PC = nv->pc;
    if (F == 0 || C == 0 || VC == 0 || VH == 0 || PC == 0 || nv->blocks < 0)
        return fail(NSG_E_FORMAT, "zero dimension in NSGW header");
    const size_t need = F * C * 9 + 4 * F + (size_t)nv->blocks * 2 * (F * F * 9 + 4 * F) +
                        PC * F + PC + VC * F + 4 * VC + VH * VC * 81 + VH + 2 * VH + 2;
    if (size != 64 + need * sizeof(float))
        return fail(NSG_E_FORMAT, "NSGW size mismatch: have %zu, header implies %zu", size,
                    64 + need * sizeof(float));
    nv->params = need;
    const float* f = (const float*)(p + 64);
    nv->stemW = f; f += F * C * 9;
    nv->stemBn = f; f += 4 * F;
    nv->w1.resize(nv->blocks); nv->bn1.resize(nv->blocks);
    nv->w2.resize(nv->blocks); nv->bn2.resize(nv->blocks);
    for (int k = 0; k < nv->blocks; ++k) {
        nv->w1[k] = f; f += F * F * 9;
        nv->bn1[k] = f; f += 4 * F;
        nv->w2[k] = f; f += F * F * 9;
        nv->bn2[k] = f; f += 4 * F;
    }
    nv->polW = f; f += PC * F;
    nv->polB = f; f += PC;
    nv->valW = f; f += VC * F;
    nv->valBn = f; f += 4 * VC;
    nv->fc1W = f; f += VH * VC * 81;
    nv->fc1B = f; f += VH;
    nv->fc2W = f; f += 2 * VH;
    nv->fc2B = f; f += 2;
    return NSG_OK;
}

// BN folding in double: w' = w * g/sqrt(v+eps), b' = beta - mean * g/sqrt(v+eps)
void foldBn(const float* bn, int n, float eps, std::vector<double>* scale, std::vector<float>* bias) {
    scale->resize(n);
    bias->resize(n);
    for (int i = 0; i < n; ++i) {
        const double g = bn[0 * n + i], b = bn[1 * n + i], m = bn[2 * n + i], v = bn[3 * n + i];
        const double s = g / std::sqrt(v + (double)eps);
        (*scale)[i] = s;
        (*bias)[i] = (float)(b - m * s);
    }
}

// kF16m8's correction operands are e4m3 copies under FIXED scales (kernels.h): x_hi as it is and
// x_lo * 2^12 must stay below 448, i.e. activations below ~224 (an f16 residual is at most 2^-11 of its
// value); beyond that the copies clamp and the error drifts from 1e-4 towards plain f16's 1e-2
// (profiles/r02/c_envelope_sweep.txt).  The bound is estimated from the WEIGHTS at load time by pushing
// a second moment through the folded layers -- uncorrelated inputs: E[y_c^2] = |w'_c|^2 E[x^2] + b'_c^2,
// a ReLU keeps half of it, a residual add sums the branches -- and taking six standard deviations of the
// widest channel.  Crude (it ignores correlations) but monotone in what matters: BN gains, weight norms.
struct MomentNet {
    double x2 = 1.0; // second moment per input value entering the next layer
    double peak = 0.0;
};
void pushConvMoment(const float* w, const double* scale, const float* bias, int cout, int cin, int taps,
                    double in2, double* out2Max, double* out2Mean) {
    double worst = 0.0, mean = 0.0;
    for (int n = 0; n < cout; ++n) {
        double ss = 0.0;
        const float* wn = w + (size_t)n * cin * taps;
        for (int k = 0; k < cin * taps; ++k) ss += (double)wn[k] * wn[k];
        const double y2 = ss * scale[n] * scale[n] * in2 + (double)bias[n] * bias[n];
        worst = std::max(worst, y2);
        mean += y2;
    }
    *out2Max = worst;
    *out2Mean = mean / cout;
}

struct Conv3Ctx { const float* w; const double* scale; int cin; };
float conv3Get(const void* c, int n, int k, int tap) {
    const Conv3Ctx* x = (const Conv3Ctx*)c;
    return (float)((double)x->w[((size_t)n * x->cin + k) * 9 + tap] * x->scale[n]);
}
struct HeadsCtx { const float* valW; const double* valScale; const float* polW; int F, vc, pc; };
float headsGet(const void* c, int n, int k, int) {
    const HeadsCtx* x = (const HeadsCtx*)c;
    if (n < x->vc) return (float)((double)x->valW[(size_t)n * x->F + k] * x->valScale[n]);
    if (n < x->vc + x->pc) return x->polW[(size_t)(n - x->vc) * x->F + k];
    return 0.f;
}
struct Fc1Ctx { const float* w; int vc; };
float fc1Get(const void* c, int n, int k, int) {
    const Fc1Ctx* x = (const Fc1Ctx*)c;
    const int sq = k / x->vc, ch = k - sq * x->vc; // device K index = sq*VC + c
    return x->w[(size_t)n * x->vc * 81 + (size_t)ch * 81 + sq];
}

constexpr int kMaxEventPairs = 1024;

} // namespace

struct nsg_evaluator {
    int gpu = 0;
    int batchMax = 0;
    int numChannels = 0;
    int prec = NSG_PRECISION_FP32;
    bool loaded = false;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop{};

    // network dims
    int F = 0, blocks = 0, vc = 0, vh = 0, cpad = 0, headsCout = 0, fc1K = 0;
    uint64_t params = 0;

    // device buffers (sized for batchMax rounded up to whole workgroups)
    DevBuf input;         // bitboards  [B][C][16 B]          (trt.cc:57-58)
    DevBuf planes;        // trunk input [Bpad][81][cpad] T   (trt.cc:59-60, other layout)
    DevBuf act[3];        // trunk activations [Bpad][81][F] T
    DevBuf policy;        // [B][2187] f32                    (trt.cc:61-62)
    DevBuf value, draw;   // [B] f32                          (trt.cc:63-66)
    DevBuf moveIdx, moveOff, gathered; // legal-move gather (allocated on first use)
    DevBuf vfeat;         // [B][fc1K] T
    DevBuf hidden;        // [K slices][B][VH] f32 partial sums of the value MLP's first layer
    DevBuf scratch;       // debug read-back

    std::shared_ptr<NetWeights> W; // shared by the evaluators of one device (nsg_load_shared)
    int lastTrunkPrec = -1; // precision the most recent forward ran its trunk in
    // host statistics (mcts::Statistics evaluationCount / batchSizeAccumulated, statistics.h:74-98)
    uint64_t statBatches = 0, statPositions = 0;
    DevBuf stamps;           // diagnostic builds: per-layer, per-workgroup cycle stamps
    DevBuf trunkLayers;      // persistent-trunk layer list (stem + 2 per block)
    int trunkLayerCount = 0;
    // One persistent launch for all 3x3 layers (a workgroup owns its boards through every layer: no grid barrier, no
    // launch boundary, workgroups drift apart instead of hitting memory in lock-step).  -1 (default): taken by a
    // kF16m6 evaluator where every tile of the batch is resident at once AND the per-layer alternatives measured
    // slower -- on 256 CUs 144..256 boards (one-board tiles; +13 % at 160-176 where it replaces the 128 + rest
    // two-part batch, +7 % at 192-256) and 384..512 boards (two-board tiles: +6-7 % on game positions, +3-4 % on the
    // benchmark's replicated initial position); with more tiles than CUs the second round waits for 41 layers of the
    // first (-35 %), 257..383 boards keep the two-part batches (profiles/r04/j_persistent_trunk_vs_per_layer_by_batch.txt).
    // NSG_TRUNK_KERNEL=1 forces it wherever the plan allows, =0 switches it off.  (Rounds 2-3, before the loop read its
    // weights and tiles through buffer descriptors: +1.6-2.4 % at 512, -3..-7 % at 160-192; r1: +-0 % kF16m8, -6 % kF16x3.)
    int useTrunkKernel = -1;

    void* trunkOut = nullptr; // which act[] holds the trunk output of the last forward

    // Team trunk (kernels/team_trunk.hip): up to sixteen boards, every 3x3 layer in one persistent launch, kF16x3
    // arithmetic.  NSG_TEAM_TRUNK=0 switches it off.  One team launch per DEVICE at a time (teamToken below).
    DevBuf teamLayers;       // nsg::TeamLayer list (stem + 2 per block) on the kF16x3 copy of the trunk
    DevBuf teamSets;         // 2 hand-off sets x 4 images x teamBoards boards (nsg::TeamHandoff)
    int teamBoards = 0;      // boards an image holds: min(batchMax, kTeamMaxBoards)
    int teamSet = 0;         // the set the next launch writes
    int teamDirty[2] = {0, 0}; // boards of each set that do not hold the sentinel
    int teamLayerCount = 0;
    int teamEnabled = 1;
    int teamForceRowGroups = 0; // NSG_TEAM_MEMBERS / 16, read when the evaluator is created (0: as many as fit)
    int teamMaxBatch = nsg::kTeamMaxBoards; // NSG_TEAM_MAX_BATCH, likewise
    int teamLastMembers = 0; // workgroups per board of the most recent team launch
    uint64_t teamFallbacks = 0; // team launches that gave up and were re-run on the per-layer kernels (nsg_get_team_stats)
    int teamLockedOut = 0;      // another PROCESS holds this device's team token (teamDeviceLock): per-layer kernels only
    bool teamLockHeld = false;  // this evaluator holds a reference on the process's lock of the device
    // Cooperative trunk (mfma_tile.h, coopTrunkKernel): the two-way K split of 65 ... CUs/2 boards as ONE launch whose
    // two workgroups per board hand their halves to each other.  NSG_COOP_TRUNK=0 switches it off.  Shares the team
    // trunk's device token (both need every workgroup of their grid resident), status word and recovery.
    DevBuf coopFlags;           // [batchMax][members] unsigned; values count on from launch to launch (coopFlagBase)
    unsigned coopFlagBase = 0;  // first flag value of the next cooperative launch
    int coopEnabled = 1;
    int coopForced = 0;         // NSG_COOP_TRUNK=1: wherever a plan allows it; unset: where it measured faster (enqueueForward)
    int lastPersistent = 0;     // what the most recent forward ran: 0 per-layer / persistent-without-hand-off, 1 team trunk, 2 cooperative trunk
    int coopFaultXccLaunches = 0; // NSG_COOP_FAULT_XCC_LAUNCHES (test hook), see enqueueCoop
    int teamFaultLaunches = 0;  // NSG_TEAM_FAULT_LAUNCHES (test hook): this many team launches are made ONE WORKGROUP SHORT,
                                // so that the team waits in vain, gives up and the recovery path runs
    // What the most recent compute call queued behind its forward: should the team launch of that forward give up
    // (teamRecover), the same batch is re-run on the per-layer kernels and these copies are issued again.
    struct Pending {
        int kind = 0; // 0 nothing, 1 nsg_compute_nonblocking, 2 nsg_compute_gather_nonblocking, 3 nsg_forward_resident
        size_t n = 0, total = 0;
        int softmax = 0;
        float* pol = nullptr; float* win = nullptr; float* draw = nullptr; float* vals = nullptr;
    } pending;
    hipEvent_t teamDone = nullptr; // behind this evaluator's most recent team launch (the token's next holder waits for it)
    int* teamStatusHost = nullptr; // host-mapped: raised by the kernel when a bounded spin runs out
    int* teamStatusDev = nullptr;
    bool teamLast = false;   // the most recent forward ran the team trunk

    // chains: large batches run as independent half-batch launch chains
    static constexpr int kMaxChains = 4;
    hipStream_t chainStream[kMaxChains - 1] = {};
    hipEvent_t forkEvent = nullptr;
    hipEvent_t joinEvent[kMaxChains - 1] = {};
    int numChains = 2;        // NSG_CHAINS (3 about equal, 4 slower: DESIGN.md 6)
    int chainMinBatch = 0;    // NSG_CHAIN_MIN_BATCH; 0 = more 2-board tiles than CUs
    nsg::ConvTuning tuning;   // NSG_CONV_* overrides, read at creation
    nsg::ConvPlan lastPlan{0, 0, 0};
    int lastChains = 0;
    // Staggered chains: chain c starts c/chains of a conv layer's duration after chain 0, so one
    // chain's memory bursts (tile loads, residual reads, output stores) fall into the other's matrix
    // phase instead of colliding with its bursts.  The layer time is measured on the first chained
    // forward of a given shape (HIP events around chain 0's residual trunk); results do not depend on it.
    // Off by default: measured +2.5 %, +2 % and +0.3 % at B = 512 (20x256, kF16m8) on three boxes, -2.4 % on
    // 10x192 and nothing below a full chip -- too fragile to be the default.
    int chainDelayUs = 0;     // NSG_CHAIN_DELAY_US: 0 = no stagger, -1 = measured layer time / chains, > 0 = fixed
    hipEvent_t calibEv[2] = {};
    bool calibPending = false;
    int calibKey = -1;        // launch shape (rounds of workgroups per chain launch, chains) the measurement belongs to
    int calibPendingKey = -1;
    float calibLayerUs = 0.f;

    // profiling
    bool profile = false;
    std::vector<hipEvent_t> ev; // 4 per forward: fwd begin, trunk begin, trunk end, fwd end
    int evUsed = 0;
    double trunkMs = 0, fwdMs = 0;
    uint64_t trunkLaunches = 0, forwards = 0;
    int pendingTrunkLaunchesPerFwd = 0;
};

namespace {

int bind(nsg_evaluator* ev) {
    NSG_HIP(hipSetDevice(ev->gpu));
    return NSG_OK;
}

int drainProfile(nsg_evaluator* ev) {
    if (ev->evUsed == 0) return NSG_OK;
    NSG_HIP(hipStreamSynchronize(ev->stream));
    for (int i = 0; i < ev->evUsed; i += 4) {
        float a = 0, b = 0;
        NSG_HIP(hipEventElapsedTime(&a, ev->ev[i + 1], ev->ev[i + 2]));
        NSG_HIP(hipEventElapsedTime(&b, ev->ev[i + 0], ev->ev[i + 3]));
        ev->trunkMs += a;
        ev->fwdMs += b;
        ev->trunkLaunches += (uint64_t)ev->pendingTrunkLaunchesPerFwd;
        ev->forwards += 1;
    }
    ev->evUsed = 0;
    return NSG_OK;
}

// Host half of a layer upload: picks the power-of-two weight scale and packs the fragment
// records.  Pure function of its arguments (runs on worker threads at load time).
void packLayer(nsg::WeightGetter get, const void* ctx, int taps, int kReal, int kdim, int cout,
               int coutReal, int prec, std::vector<unsigned char>* host, float* accScale) {
    host->assign(nsg::tileWeightRecords(taps, kdim, cout, prec) * 16, 0);
    float scale = 1.f;
    if (prec == nsg::kF16x3 || nsg::isMx(prec)) {
        // power-of-two scale putting the largest |w| in [2^8, 2^9): hi and lo of every
        // weight that matters stay in f16's normal range; undone exactly by accScale
        float maxAbs = 0.f;
        for (int n = 0; n < coutReal; ++n)
            for (int k = 0; k < kReal; ++k)
                for (int t = 0; t < taps; ++t) maxAbs = std::fmax(maxAbs, std::fabs(get(ctx, n, k, t)));
        if (maxAbs > 0.f && std::isfinite(maxAbs)) {
            int e = 0;
            std::frexp(maxAbs, &e); // maxAbs = m * 2^e, m in [0.5, 1)
            scale = std::ldexp(1.0f, 9 - e);
        }
    }
    *accScale = 1.0f / scale;
    nsg::packTileWeights(get, ctx, taps, kReal, kdim, cout, prec, scale, host->data());
}

// Device half: allocation and copies (caller's thread, bound to the evaluator's device).
int commitLayer(const std::vector<unsigned char>& host, float accScale, int kdim, int cout,
                const std::vector<float>& bias, ConvLayer* L) {
    L->accScale = accScale;
    int rc = L->w.alloc(host.size(), false);
    if (rc) return rc;
    NSG_HIP(hipMemcpy(L->w.p, host.data(), host.size(), hipMemcpyHostToDevice));
    std::vector<float> b(cout, 0.f);
    for (size_t i = 0; i < bias.size() && i < (size_t)cout; ++i) b[i] = bias[i];
    rc = L->bias.alloc((size_t)cout * 4, false);
    if (rc) return rc;
    NSG_HIP(hipMemcpy(L->bias.p, b.data(), (size_t)cout * 4, hipMemcpyHostToDevice));
    L->cin = kdim;
    return NSG_OK;
}

int uploadLayer(nsg::WeightGetter get, const void* ctx, int taps, int kReal, int kdim,
                int cout, int coutReal, int prec, const std::vector<float>& bias, ConvLayer* L) {
    std::vector<unsigned char> host;
    float accScale = 1.f;
    packLayer(get, ctx, taps, kReal, kdim, cout, coutReal, prec, &host, &accScale);
    return commitLayer(host, accScale, kdim, cout, bias, L);
}

int roundUp(int a, int b) { return (a + b - 1) / b * b; }

// One wave that spins for `ticks` of the 100 MHz constant clock: phase-shifts a chain.
__global__ void delayKernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// Two team launches on one device could each hold CUs that the other's not-yet-scheduled members need.  One token
// per device: an evaluator launches a team trunk if it holds the token already (its launches are ordered on its own
// stream); otherwise its stream first waits -- on the device, hipStreamWaitEvent -- for the event behind the holder's
// most recent team launch, and it takes the token.  Waiting -- not falling back to the per-layer kernels -- keeps the
// arithmetic of a batch independent of timing: a self-play engine's two executors then produce the same bits in
// every run.  The mutex is held from the hand-over of the token until the new holder's launch AND its event are in
// its stream (TeamTokenGuard around enqueueTeam): released in between, the old holder could come back, find nothing
// to wait for yet, take the token back and launch beside the new one -- two engine threads of one self-play process
// did exactly that, and both launches timed out.
std::mutex gTeamMutex;
nsg_evaluator* gTeamOwner[64] = {};
struct TeamTokenGuard {
    std::unique_lock<std::mutex> lock;
    bool held = false;
    explicit TeamTokenGuard(nsg_evaluator* ev) : lock(gTeamMutex) {
        nsg_evaluator*& owner = gTeamOwner[ev->gpu & 63];
        if (owner != ev && owner != nullptr && owner->teamDone &&
            hipStreamWaitEvent(ev->stream, owner->teamDone, 0) != hipSuccess)
            return;
        owner = ev;
        held = true;
    }
};
void releaseTeamToken(nsg_evaluator* ev) {
    std::lock_guard<std::mutex> lock(gTeamMutex);
    if (gTeamOwner[ev->gpu & 63] == ev) gTeamOwner[ev->gpu & 63] = nullptr;
}

// ... and one PROCESS per device: two processes' team launches would starve each other the same way, and no stream
// event reaches across processes.  The first evaluator of a process that builds a team layer list for a device takes
// an advisory lock (flock, released when the process ends) on a file named after the device's PCI bus id; a process
// that finds the lock taken keeps to the per-layer kernels on that device.  (A file system the other process does
// not share -- another container -- is not covered: there the bounded spins and teamRecover are what is left.)
// The lock is counted: the last evaluator of the process that used it gives it back (nsg_destroy).
// NSG_TEAM_LOCK_DIR names the directory of the lock files (default /dev/shm, then /tmp) -- for processes that share a
// device but not those directories.  Returns false only when ANOTHER process holds the lock.
struct TeamDeviceLocks {
    std::mutex m;
    int fd[64];
    int refs[64];
    TeamDeviceLocks() { for (int i = 0; i < 64; ++i) { fd[i] = -1; refs[i] = 0; } }
};
TeamDeviceLocks& teamLocks() {
    static TeamDeviceLocks t;
    return t;
}
bool teamDeviceLock(int gpu) {
    TeamDeviceLocks& T = teamLocks();
    std::lock_guard<std::mutex> lock(T.m);
    const int g = gpu & 63;
    if (T.refs[g] > 0) { ++T.refs[g]; return true; } // this process holds it (or found nothing to lock: fd -1)
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, sizeof(bus), gpu) != hipSuccess || !bus[0]) snprintf(bus, sizeof(bus), "gpu%d", gpu);
    for (char* c = bus; *c; ++c)
        if (*c == ':' || *c == '/' || *c == '.') *c = '_';
    const char* envDir = getenv("NSG_TEAM_LOCK_DIR");
    const char* dirs[2] = {envDir && *envDir ? envDir : "/dev/shm", envDir && *envDir ? nullptr : "/tmp"};
    for (const char* dir : dirs) {
        if (!dir) continue;
        char path[512];
        snprintf(path, sizeof(path), "%s/nsg_team_token_%s.lock", dir, bus);
        const mode_t old = umask(0);
        const int f = open(path, O_CREAT | O_RDWR | O_CLOEXEC, 0666);
        umask(old);
        if (f < 0) continue;
        if (flock(f, LOCK_EX | LOCK_NB) == 0) { T.fd[g] = f; T.refs[g] = 1; return true; }
        const bool taken = errno == EWOULDBLOCK;
        close(f);
        if (taken) return false;
    }
    T.fd[g] = -1; // no lock file could be opened: nothing to coordinate through (see above)
    T.refs[g] = 1;
    return true;
}
void teamDeviceUnlock(int gpu) {
    TeamDeviceLocks& T = teamLocks();
    std::lock_guard<std::mutex> lock(T.m);
    const int g = gpu & 63;
    if (T.refs[g] > 0 && --T.refs[g] == 0 && T.fd[g] >= 0) {
        close(T.fd[g]); // (closing the descriptor gives the flock back)
        T.fd[g] = -1;
    }
}

// Workgroups per board of a team launch of B boards on this evaluator's device; 0: the batch does not run as a team.
int teamMembersFor(const nsg_evaluator* ev, int B) {
    if (ev->teamLayerCount <= 0 || !ev->teamEnabled || B > ev->teamMaxBatch) return 0;
    return nsg::teamMembers(B, ev->F, ev->prop.multiProcessorCount, ev->teamForceRowGroups);
}

// The whole forward of a batch of at most kTeamMaxBoards boards with the team trunk.
int enqueueTeam(nsg_evaluator* ev, int B, int members, hipStream_t s, hipEvent_t trunkBegin, hipEvent_t trunkEnd) {
    const int prec = nsg::kF16x3;
    ev->lastTrunkPrec = prec;
    {   // (the feature bitboards are decoded by the first layer of the team launch itself)
        Range r("nsg.trunk");
        if (trunkBegin) NSG_HIP(hipEventRecord(trunkBegin, s));
        nsg::TeamHandoff ho;
        ho.imageStride = (size_t)ev->teamBoards * 81 * 1024;
        const size_t setBytes = 4 * ho.imageStride;
        ho.set = (unsigned char*)ev->teamSets.p + (size_t)ev->teamSet * setBytes;
        ho.other = (unsigned char*)ev->teamSets.p + (size_t)(1 - ev->teamSet) * setBytes;
        ho.cleanBoards = ev->teamDirty[1 - ev->teamSet];
        ho.bits = ev->input.p;
        ho.bitChannels = ev->numChannels;
        const int shortBy = ev->teamFaultLaunches > 0 ? 1 : 0;
        if (shortBy) --ev->teamFaultLaunches;
        NSG_HIP(nsg::launchTeamTrunk((const nsg::TeamLayer*)ev->teamLayers.p, ev->teamLayerCount, B, ev->F, members, ho,
                                     ev->teamStatusDev, s, shortBy));
        ev->teamLastMembers = members;
        NSG_HIP(hipEventRecord(ev->teamDone, s)); // (under the token's mutex: enqueueForward)
        ev->teamDirty[1 - ev->teamSet] = 0;
        ev->teamDirty[ev->teamSet] = B;
        ev->teamSet = 1 - ev->teamSet;
        if (trunkEnd) NSG_HIP(hipEventRecord(trunkEnd, s));
    }
    void* x = (ev->blocks % 2 == 1) ? ev->act[2].p : ev->act[0].p; // the buffer rotation of the layer list
    ev->trunkOut = x;
    Range headsRange("nsg.heads");
    NSG_HIP(nsg::launchHeads(x, ev->W->heads.w.p, (const float*)ev->W->heads.bias.p, (float*)ev->policy.p, ev->vfeat.p, B,
                             ev->F, ev->headsCout, ev->vc, ev->fc1K, ev->W->heads.accScale, prec, s));
    const size_t partStride = (size_t)ev->batchMax * ev->vh;
    NSG_HIP(nsg::launchDense(ev->vfeat.p, ev->W->fc1.w.p, (const float*)ev->W->fc1.bias.p, (float*)ev->hidden.p, B,
                             ev->fc1K, ev->vh, partStride, ev->W->fc1.accScale, prec, s));
    NSG_HIP(nsg::launchValueOut((const float*)ev->hidden.p, (const float*)ev->W->fc1.bias.p, nsg::denseSplits(ev->fc1K, prec),
                                partStride, (const float*)ev->W->fc2W.p, (const float*)ev->W->fc2B.p, (float*)ev->value.p,
                                (float*)ev->draw.p, B, ev->vh, s));
    return NSG_OK;
}

// The value / policy heads behind a trunk whose output is `x` (the tail of every forward).
int enqueueHeads(nsg_evaluator* ev, void* x, int B, int hprec, hipStream_t s) {
    Range headsRange("nsg.heads");
    NSG_HIP(nsg::launchHeads(x, ev->W->heads.w.p, (const float*)ev->W->heads.bias.p, (float*)ev->policy.p, ev->vfeat.p, B,
                             ev->F, ev->headsCout, ev->vc, ev->fc1K, ev->W->heads.accScale, hprec, s));
    const size_t partStride = (size_t)ev->batchMax * ev->vh;
    NSG_HIP(nsg::launchDense(ev->vfeat.p, ev->W->fc1.w.p, (const float*)ev->W->fc1.bias.p, (float*)ev->hidden.p, B,
                             ev->fc1K, ev->vh, partStride, ev->W->fc1.accScale, hprec, s));
    NSG_HIP(nsg::launchValueOut((const float*)ev->hidden.p, (const float*)ev->W->fc1.bias.p, nsg::denseSplits(ev->fc1K, hprec),
                                partStride, (const float*)ev->W->fc2W.p, (const float*)ev->W->fc2B.p, (float*)ev->value.p,
                                (float*)ev->draw.p, B, ev->vh, s));
    return NSG_OK;
}

// The whole forward of a mid batch with the cooperative trunk (under the device's token: enqueueForward).
int enqueueCoop(nsg_evaluator* ev, int B, const nsg::ConvPlan& plan, hipStream_t s, hipEvent_t trunkBegin, hipEvent_t trunkEnd) {
    const int prec = ev->prec;
    ev->lastTrunkPrec = prec;
    {
        Range r("nsg.planes");
        NSG_HIP(nsg::launchExtractBitsAct(ev->planes.p, (const uint64_t*)ev->input.p, B, ev->numChannels, ev->cpad, prec, s));
    }
    {
        Range r("nsg.trunk");
        const int members = nsg::coopMembers(ev->F, plan);
        bool stemFirst = false;
        (void)nsg::canRunCoopTrunk(ev->F, ev->cpad, prec, plan, &stemFirst);
        // flag values count on across launches (24 bits; the array is cleared when they run out): no memset per forward
        if (ev->coopFlagBase + 256u >= (1u << 24)) {
            NSG_HIP(hipMemsetAsync(ev->coopFlags.p, 0, ev->coopFlags.bytes, s));
            ev->coopFlagBase = 0;
        }
        const unsigned flagBase = ev->coopFlagBase;
        ev->coopFlagBase += 128; // (more than any net's 3x3 layers: 1 + 2 x 40)
        (void)members;
        if (stemFirst) // the stem has fewer chunk pairs than the plan splits K by: its own launch, the plan's stem kernel
            NSG_HIP(nsg::launchConv3x3(ev->planes.p, ev->W->stem.w.p, (const float*)ev->W->stem.bias.p, nullptr, ev->act[0].p, B,
                                       ev->cpad, ev->F, 1, ev->W->stem.accScale, prec, plan, s, nullptr, false));
        const int l0 = stemFirst ? 1 : 0;
        if (trunkBegin) NSG_HIP(hipEventRecord(trunkBegin, s));
        // (test hook NSG_TEAM_FAULT_LAUNCHES: the last board's second member leaves at once and never publishes)
        // (NSG_COOP_FAULT_XCC_LAUNCHES: every board's second member claims another XCD, what a placement the hand-off
        // cannot use looks like)
        int faultBoard = ev->teamFaultLaunches > 0 ? B - 1 : -1;
        if (faultBoard >= 0) --ev->teamFaultLaunches;
        else if (ev->coopFaultXccLaunches > 0) { faultBoard = -2; --ev->coopFaultXccLaunches; }
        NSG_HIP(nsg::launchCoopTrunk((const unsigned char*)ev->trunkLayers.p + (size_t)l0 * nsg::trunkLayerBytes(),
                                     ev->trunkLayerCount - l0, B, ev->F, prec, plan, (unsigned*)ev->coopFlags.p, flagBase,
                                     ev->teamStatusDev, s, faultBoard));
        NSG_HIP(hipEventRecord(ev->teamDone, s));
        if (trunkEnd) NSG_HIP(hipEventRecord(trunkEnd, s));
    }
    void* x = (ev->blocks % 2 == 1) ? ev->act[2].p : ev->act[0].p; // the buffer rotation of the layer list
    ev->trunkOut = x;
    return enqueueHeads(ev, x, B, nsg::headPrecision(prec), s);
}

// One chain = the whole forward for boards [off, off + count) on stream `s`.
int enqueueChain(nsg_evaluator* ev, int off, int count, const nsg::ConvPlan& plan, hipStream_t s,
                 bool stampsOk, hipEvent_t trunkBegin = nullptr, hipEvent_t trunkEnd = nullptr) {
    // kF16m8 runs full tiles only; smaller launch plans use the kF16x3 copy of the trunk
    const bool x3Fallback = nsg::isMx(ev->prec) && (plan.nfrag != 4 || ev->W->outsideM8Window);
    const int prec = x3Fallback ? (int)nsg::kF16x3 : ev->prec;
    const ConvLayer& stem = x3Fallback ? ev->W->stemX3 : ev->W->stem;
    const std::vector<ConvLayer>& conv1 = x3Fallback ? ev->W->conv1X3 : ev->W->conv1;
    const std::vector<ConvLayer>& conv2 = x3Fallback ? ev->W->conv2X3 : ev->W->conv2;
    if (off == 0) ev->lastTrunkPrec = prec;
    const size_t es = (size_t)nsg::elemSize(prec);
    auto act = [&](void* base, size_t rowElems) { return (void*)((unsigned char*)base + (size_t)off * rowElems * es); };
    const uint64_t* input = (const uint64_t*)ev->input.p + (size_t)off * ev->numChannels * 2;
    void* planes = act(ev->planes.p, (size_t)81 * ev->cpad);
    // feature planes (replaces cuda::extractBits, trt.cc:255-258)
    {
        Range r("nsg.planes");
        NSG_HIP(nsg::launchExtractBitsAct(planes, input, count, ev->numChannels, ev->cpad, prec, s));
    }
    Range trunkRange("nsg.trunk");
    // stem + residual trunk
    void* x = act(ev->act[0].p, (size_t)81 * ev->F);
    void* y = act(ev->act[1].p, (size_t)81 * ev->F);
    void* z = act(ev->act[2].p, (size_t)81 * ev->F);
    const int hprec = nsg::headPrecision(prec); // kF16m8: the last trunk layer writes the kF16x3 layout
    NSG_HIP(nsg::launchConv3x3(planes, stem.w.p, (const float*)stem.bias.p, nullptr, x, count,
                               ev->cpad, ev->F, 1, stem.accScale, prec, plan, s, nullptr,
                               hprec != prec && ev->blocks == 0));
    if (trunkBegin) NSG_HIP(hipEventRecord(trunkBegin, s));
    for (int k = 0; k < ev->blocks; ++k) {
        unsigned long long* st1 = nullptr;
        unsigned long long* st2 = nullptr;
#ifdef NSG_DIAG_STAMPS
        if (ev->stamps.p && stampsOk) {
            st1 = (unsigned long long*)ev->stamps.p + (size_t)(2 * k) * 4096 * 8;
            st2 = (unsigned long long*)ev->stamps.p + (size_t)(2 * k + 1) * 4096 * 8;
        }
#else
        (void)stampsOk;
#endif
        NSG_HIP(nsg::launchConv3x3(x, conv1[k].w.p, (const float*)conv1[k].bias.p, nullptr, y, count,
                                   ev->F, ev->F, 1, conv1[k].accScale, prec, plan, s, st1));
        NSG_HIP(nsg::launchConv3x3(y, conv2[k].w.p, (const float*)conv2[k].bias.p, x, z, count,
                                   ev->F, ev->F, 1, conv2[k].accScale, prec, plan, s, st2,
                                   hprec != prec && k == ev->blocks - 1));
        void* t = x; x = z; z = t;
    }
    if (trunkEnd) NSG_HIP(hipEventRecord(trunkEnd, s));
    if (off == 0) ev->trunkOut = x;
    if (trunkRange.on) { roctx().pop(); trunkRange.on = false; }
    Range headsRange("nsg.heads");
    // heads
    float* policy = (float*)ev->policy.p + (size_t)off * NSG_MOVE_INDEX_MAX;
    void* vfeat = act(ev->vfeat.p, (size_t)ev->fc1K);
    float* hidden = (float*)ev->hidden.p + (size_t)off * ev->vh;
    NSG_HIP(nsg::launchHeads(x, ev->W->heads.w.p, (const float*)ev->W->heads.bias.p, policy, vfeat, count, ev->F,
                             ev->headsCout, ev->vc, ev->fc1K, ev->W->heads.accScale, hprec, s));
    const size_t partStride = (size_t)ev->batchMax * ev->vh; // floats between the K slices' partial sums
    NSG_HIP(nsg::launchDense(vfeat, ev->W->fc1.w.p, (const float*)ev->W->fc1.bias.p, hidden, count, ev->fc1K,
                             ev->vh, partStride, ev->W->fc1.accScale, hprec, s));
    NSG_HIP(nsg::launchValueOut(hidden, (const float*)ev->W->fc1.bias.p, nsg::denseSplits(ev->fc1K, hprec), partStride,
                                (const float*)ev->W->fc2W.p, (const float*)ev->W->fc2B.p,
                                (float*)ev->value.p + off, (float*)ev->draw.p + off, count, ev->vh, s));
    return NSG_OK;
}

// The tile plan of a batch of B boards on this evaluator (precision, channel count, CU count, tuning).
nsg::ConvPlan planForBatch(nsg_evaluator* ev, int B) {
    nsg::ConvPlan plan = nsg::chooseConvPlan(B, ev->F, ev->prop.multiProcessorCount, ev->tuning);
    // kF16m8 keeps four image buffers in LDS (125 KB for two boards): a CU holds one two-board
    // workgroup, so where the channel count only allows two-wave workgroups (F = 384) one-board
    // tiles (74 KB) keep all four SIMDs busy with two workgroups per CU
    // (an F16M8 evaluator whose network left the fixed-scale window runs as kF16x3: none of the MX plans apply)
    const bool mx = nsg::isMx(ev->prec) && !ev->W->outsideM8Window;
    if (mx && plan.nfrag == 4 && plan.nwaves <= 2 && ev->tuning.nb == 0) plan.nb = 1;
    // Mid batches -- the engine's default batch of 128 and its benchmark's 60..159 -- run kF16m8
    // one-board tiles whose four waves are two row groups x two 64-channel groups (two workgroups per
    // board) wherever those fill more than half the CUs in one round of workgroups
    if (mx && plan.nfrag != 4 && ev->F % 128 == 0 && ev->tuning.nfrag == 0 &&
        ev->tuning.msplit != 1) {
        const long cus = ev->prop.multiProcessorCount;
        const long wgM8 = (long)B * (ev->F / 128);
        // (NSG_KSPLIT4_MAX_BATCH: experiment knob, read once -- largest batch that takes the four-way K split)
        static const int k4max = [] { const char* e = getenv("NSG_KSPLIT4_MAX_BATCH"); return e ? atoi(e) : -1; }();
        if (ev->F == 256 && ev->cpad == 128 && ev->tuning.msplit != 2 &&
            (k4max >= 0 ? B <= k4max : (long)B * 4 <= cus)) {
            // small batches: four workgroups per board, each one 64-channel group whose four waves take one chunk
            // pair apiece (K split by four, whole board resident in LDS)
            plan.nb = 1; plan.nfrag = 4; plan.nwaves = 4; plan.msplit = 1; plan.ksplit = 4;
            // ... and while eight workgroups per board still fit the chip in one round, the rows are split
            // over two workgroups as well (NSG_ROWSPLIT8_MAX_BATCH, read when the evaluator is created)
            // -- two, three or six row groups of three, two or one fragment: as many as fit the chip in one round
            const int k8max = ev->tuning.rowsplit8Max;
            if (k8max >= 0 ? B <= k8max : true) {
                // (three row groups = twelve workgroups per board: three boards' do not fit an XCD, so 17-21 boards have
                // no cooperative form with them; two row groups as ONE cooperative launch beat three as per-layer
                // launches by 2...9 %: profiles/r04/zn_*.  Without the cooperative trunk -- switched off, the device's
                // lock lost, after a give-up -- three row groups it is.)
                const bool coop = ev->coopEnabled && ev->prec == nsg::kF16m6 && ev->teamStatusDev && ev->coopFlags.p;
                if ((long)B * 24 <= cus) plan.msplit = 6;
                else if ((long)B * 12 <= cus && !coop) plan.msplit = 3;
                else if ((long)B * 8 <= cus) plan.msplit = 2;
            }
        } else
        if (wgM8 <= cus && wgM8 * 2 > cus) { // (measured: 1.17-1.29 ms for every B in 65..128; kF16x3 plans 1.26-1.54 ms)
            plan.nb = 1; plan.nfrag = 4; plan.nwaves = 4; plan.msplit = 2;
            // K split instead of the row split where the whole board fits in LDS (<= 256 channels) and both
            // the stem's and the trunk's chunk pairs halve evenly; NSG_CONV_MSPLIT=2 keeps the row split
            const int stemChunks = ev->cpad / 32, trunkChunks = ev->F / 32;
            if (ev->tuning.msplit != 2 && trunkChunks <= 8 && trunkChunks % 4 == 0 && stemChunks % 4 == 0 && stemChunks <= 8) {
                plan.msplit = 1;
                plan.ksplit = 2;
            }
        }
    }

    // 192 channels (BASELINE configs[1]: 10x192 at batch 64) are three chunk pairs: up to CUs/3 boards run one
    // 64-channel group per workgroup whose three waves take one pair each, all of the board's chunk tiles resident
    // (the four-way K split of the 256-channel nets, three ways), the rows over as many workgroups as fit one round.
    if (mx && ev->F == 192 && ev->cpad == 128 && ev->tuning.nb == 0 && ev->tuning.nfrag == 0 && ev->tuning.nwaves == 0 &&
        ev->tuning.msplit != 2 && ev->tuning.ksplit3 != 0) {
        const long cus = ev->prop.multiProcessorCount;
        if ((long)B * 3 <= cus) {
            plan = nsg::ConvPlan{};
            plan.nb = 1; plan.nfrag = 4; plan.nwaves = 3; plan.msplit = 1; plan.ksplit = 3;
            if (ev->tuning.msplit != 1) {
                if ((long)B * 18 <= cus) plan.msplit = 6;
                else if ((long)B * 9 <= cus) plan.msplit = 3;
                else if ((long)B * 6 <= cus) plan.msplit = 2;
            }
        }
    }

    // Mid batches, MX arithmetic: two-board tiles of ONE 64-channel group (or two) per workgroup whose waves split
    // every chunk pair's slabs between them (mfma_tile.h, OwnSeq) -- where one workgroup per (two boards, group)
    // fills more than half the CUs in one round and the four-workgroups-per-board K split does not apply.  Against
    // the one-board tiles it replaces: a workgroup streams half the weights for twice the boards, and the
    // two-board tile's edge-packed rows need 19 % fewer matrix instructions per board.  MEASURED SLOWER (round 3,
    // profiles/r03/d_*): 113k against 123k evals/s at 128 boards, 134k against 151k at 256 -- a wave's share of a
    // chunk pair is 4.2k cycles of MFMAs, and the pair's two tile round trips, its barrier and the one-slab lead of
    // the MX weight records no longer hide behind it.  Opt-in: NSG_SLAB_SPLIT=1.
    if (mx && ev->tuning.slabSplit == 1 && ev->tuning.nb == 0 && ev->tuning.nfrag == 0 && ev->tuning.nwaves == 0 &&
        ev->tuning.msplit == 0 && !(plan.ksplit == 4)) {
        const long cus = ev->prop.multiProcessorCount;
        const long tiles = (B + 1) / 2;
        for (int ss : {4, 2}) {
            if (ev->F % (256 / ss) != 0) continue;
            const long wgs = tiles * (ev->F / (256 / ss));
            if (wgs <= cus && wgs * 2 > cus) {
                plan = nsg::ConvPlan{};
                plan.nb = 2; plan.nfrag = 4; plan.nwaves = 4; plan.sslab = ss;
                break;
            }
        }
    }

    return plan;
}

int enqueueForward(nsg_evaluator* ev, size_t n) {
    const int B = (int)n;
    ++ev->statBatches;
    ev->statPositions += n;
    hipStream_t s = ev->stream;
    // the tile plan is chosen for the whole batch: all chains run concurrently
    nsg::ConvPlan plan = planForBatch(ev, B);
    const bool mx = nsg::isMx(ev->prec) && !ev->W->outsideM8Window;
    const bool prof = ev->profile;
    if (prof && ev->evUsed + 4 > (int)ev->ev.size()) {
        int rc = drainProfile(ev);
        if (rc) return rc;
    }
    hipEvent_t* e = prof ? &ev->ev[ev->evUsed] : nullptr;
    if (prof) NSG_HIP(hipEventRecord(e[0], s));

    // The smallest batches: every 3x3 layer in one persistent launch, a board per team of 16-96 workgroups
    // (kernels/team_trunk.hip), when no tuning override asks for a particular per-layer plan and no other evaluator
    // has a team launch in flight on this device.
    ev->teamLast = false;
    ev->lastPersistent = 0;
    // (a give-up word still raised here belongs to a forward nobody waited for: its batch is gone, the team path is not
    // taken again)
    if (ev->teamStatusHost && *ev->teamStatusHost != 0) {
        *ev->teamStatusHost = 0;
        ev->teamEnabled = 0;
        ev->coopEnabled = 0;
        ++ev->teamFallbacks;
        releaseTeamToken(ev);
    }
    const int members = (ev->tuning.nb == 0 && ev->tuning.nfrag == 0 && ev->tuning.nwaves == 0 && ev->tuning.msplit == 0 &&
                         ev->useTrunkKernel != 1) ? teamMembersFor(ev, B) : 0;
    if (members > 0) {
        int rc;
        {
            TeamTokenGuard token(ev);
            if (!token.held) return fail(NSG_E_HIP, "team trunk: hipStreamWaitEvent on the device's other team launch failed");
            rc = enqueueTeam(ev, B, members, s, prof ? e[1] : nullptr, prof ? e[2] : nullptr);
        }
        if (rc) return rc;
        ev->teamLast = true;
        ev->lastPersistent = 1;
        plan = nsg::ConvPlan{};
        plan.nb = 1; plan.nfrag = 1; plan.nwaves = 8; plan.ksplit = 8; // 16 weight fragments x 2 or 6 row groups per board, K over 8 waves
        plan.msplit = members / (ev->F / 16); // row groups
        ev->lastPlan = plan;
        ev->lastChains = 1;
        if (prof) {
            NSG_HIP(hipEventRecord(e[3], s));
            ev->evUsed += 4;
            ev->pendingTrunkLaunchesPerFwd = 1;
        }
        return NSG_OK;
    }

    // Mid batches on the two-way K split: the cooperative trunk, when every workgroup of its grid is resident at once
    // (same token as the team trunk: one such launch per device at a time)
    if (ev->coopEnabled && ev->teamStatusDev && ev->coopFlags.p && ev->tuning.nb == 0 && ev->tuning.nfrag == 0 &&
        ev->tuning.nwaves == 0 && ev->tuning.msplit == 0 && ev->useTrunkKernel != 1 && mx && ev->chainMinBatch <= 0 &&
        ev->chainDelayUs == 0 && nsg::canRunCoopTrunk(ev->F, ev->cpad, ev->prec, plan) &&
        nsg::coopFits(B, nsg::coopMembers(ev->F, plan), ev->prop.multiProcessorCount) &&
        ev->trunkLayerCount < 127 && // (a launch's flag values fit the 128 the base advances by)
        // measured: 256 channels with up to eight members per board +1...8 % (24-128 boards;
        // profiles/r04/o_cooperative_trunk_all_k_split_plans_sweep.txt; twelve members, 17-21 boards, do not fit an
        // XCD's CUs three boards at a time); 192 channels +1...9 % (17-80 boards) since the hand-off stays in the XCD's L2
        // (profiles/r04/zd_cooperative_trunk_192_channels.txt; -2...-8 % with write-through stores)
        (ev->coopForced || (ev->F == 256 && nsg::coopMembers(ev->F, plan) <= 8) || ev->F == 192)) {
        int rc;
        {
            TeamTokenGuard token(ev);
            if (!token.held) return fail(NSG_E_HIP, "cooperative trunk: hipStreamWaitEvent on the device's other persistent launch failed");
            rc = enqueueCoop(ev, B, plan, s, prof ? e[1] : nullptr, prof ? e[2] : nullptr);
        }
        if (rc) return rc;
        ev->teamLast = true;
        ev->lastPersistent = 2;
        ev->lastPlan = plan;
        ev->lastChains = 1;
        if (prof) {
            NSG_HIP(hipEventRecord(e[3], s));
            ev->evUsed += 4;
            ev->pendingTrunkLaunchesPerFwd = 1;
        }
        return NSG_OK;
    }

    // A batch with more tiles than CUs runs as two independent chains of half-batch
    // launches on separate streams: boards never interact, so chain A's layer l+1 may
    // start while chain B is still in layer l.  The partly filled last round of
    // workgroups of one chain's launch is topped up by the other chain's launch instead
    // of idling the chip (B=640: 82.7k -> 111.6k evals/s; B=1024: 113.8k -> 119.7k).
    // With NSG_CHAIN_DELAY_US set, two staggered chains (below) also run when all tiles are resident at
    // once on more than half the chip (B = 512 on 256 CUs; not the f32 kernel, which is matrix-bound at
    // 0.85 of its peak and loses 2 %).  With more tiles than CUs the chains compete for CUs and topping-up
    // matters more than phase: those always run unstaggered (staggered B = 640 loses 2 %).
    const int cus = ev->prop.multiProcessorCount;
    const int tiles = ((B + 1) / 2) * std::max(1, ev->F / (plan.nwaves * plan.nfrag * 16)); // of a 2-board plan
    const bool oneRound = tiles <= cus;
    int chains = 1;
    if (plan.nb == 2) {
        if (ev->chainMinBatch > 0) chains = B >= ev->chainMinBatch ? ev->numChains : 1;
        else if (!oneRound) chains = ev->numChains;
        else if (2 * tiles > cus && ev->chainDelayUs != 0 && ev->prec != nsg::kFp32) chains = std::min(ev->numChains, 2);
    }
    const bool stagger = oneRound && chains > 1;
    bool trunkWanted = ev->useTrunkKernel == 1;
    if (ev->useTrunkKernel < 0 && ev->prec == nsg::kF16m6 && plan.nfrag == 4 && ev->tuning.nb == 0 && ev->tuning.nfrag == 0 &&
        ev->tuning.nwaves == 0 && ev->tuning.msplit == 0 && ev->chainMinBatch <= 0 && ev->chainDelayUs == 0) {
        const long t = (B + plan.nb - 1) / plan.nb; // tiles = workgroups of the persistent launch
        trunkWanted = t <= cus && (plan.nb == 1 ? t * 16 >= (long)cus * 9 : t * 4 >= (long)cus * 3);
    }
    const bool trunkKernel = trunkWanted && !ev->W->outsideM8Window && nsg::canRunTrunk(ev->F, plan);
    if (trunkKernel) chains = 1;
    const int per = ((B + chains - 1) / chains + 1) / 2 * 2; // boards per chain, whole 2-board tiles
    ev->lastPlan = plan;
    ev->lastChains = chains;

    // Two-part batches.  Between the sizes whose plans fill the chip exactly, one plan for the whole batch leaves
    // CUs idle (B = CUs/2 + 1 .. CUs as one-board tiles: one workgroup per board) or pays a two-board tile for a
    // half-empty chip (B = CUs + 1 ..).  Boards never interact, so such a batch runs as a FULL part with the plan of
    // the size below (CUs/2 boards, two-way K split; CUs boards, one-board tiles) plus the remainder with ITS plan
    // (up to CUs/4 boards: the K split by four), on two streams like the half-batch chains above.
    struct Part { int off, count; nsg::ConvPlan plan; };
    Part parts[3];
    int nParts = 0;
    if (mx && !trunkKernel && ev->numChains >= 2 && ev->tuning.splitBatch != 0 && ev->F == 256 && ev->cpad == 128 &&
        ev->tuning.nb == 0 && ev->tuning.nfrag == 0 && ev->tuning.nwaves == 0 && ev->tuning.msplit == 0 &&
        ev->chainMinBatch <= 0 && ev->chainDelayUs == 0) {
        // (measured on 256 CUs, profiles/r02/n_ab_two_part_batches.txt: 129 +10.6 %, 144 +7.7 %, 168 +5.7 %, 176 +2.5 %,
        // 192 -5.9 %; 257 +52 %, 288 +37 %, 320 +27 %, 352 +18 %, 368 +14 %, 384 -7 %: from 3/2 of the CUs on
        // the two-board tiles cover three quarters of the chip and win)
        int off = 0, rem = B;
        auto take = [&](int count) { parts[nParts++] = Part{off, count, planForBatch(ev, count)}; off += count; rem -= count; };
        // more two-board tiles than CUs: one full chip of them first, the rest by the rules below (instead of two
        // half-batch chains; only just above 2 CUs boards: 513-576 +2-3.5 %, from 640 on the two chains are 3-11 %
        // faster, profiles/r02/n_ab_two_part_batches.txt)
        if (rem > 2 * cus && rem <= ev->tuning.splitBatchMax3 * cus / 4) take(2 * cus);
        if (rem > cus && rem < cus + cus / 2) take(cus);
        else if (rem > cus / 2 && rem <= cus / 2 + 3 * cus / 16) take(cus / 2);
        if (nParts > 0 && rem > 0) take(rem);
        if (nParts < 2) nParts = 0;
    }
    if (nParts > 0) {
        ev->lastPlan = parts[0].plan;
        ev->lastChains = nParts;
        NSG_HIP(hipEventRecord(ev->forkEvent, s));
        if (prof) NSG_HIP(hipEventRecord(e[1], s));
        for (int c = 0; c < nParts; ++c) {
            hipStream_t cs = (c == 0) ? s : ev->chainStream[c - 1];
            if (c > 0) NSG_HIP(hipStreamWaitEvent(cs, ev->forkEvent, 0));
            int rc = enqueueChain(ev, parts[c].off, parts[c].count, parts[c].plan, cs, c == 0);
            if (rc) return rc;
            if (c > 0) {
                NSG_HIP(hipEventRecord(ev->joinEvent[c - 1], cs));
                NSG_HIP(hipStreamWaitEvent(s, ev->joinEvent[c - 1], 0));
            }
        }
        if (prof) NSG_HIP(hipEventRecord(e[2], s));
    } else
    if (chains == 1 && trunkKernel) {
        // one persistent launch for all 2N+1 3x3 layers (measured slower; NSG_TRUNK_KERNEL=1)
        const int prec = ev->prec;
        NSG_HIP(nsg::launchExtractBitsAct(ev->planes.p, (const uint64_t*)ev->input.p, B, ev->numChannels,
                                          ev->cpad, prec, s));
        if (prof) NSG_HIP(hipEventRecord(e[1], s));
        NSG_HIP(nsg::launchTrunk(ev->trunkLayers.p, ev->trunkLayerCount, B, prec, plan, s));
        if (prof) NSG_HIP(hipEventRecord(e[2], s));
        void* x = (ev->blocks % 2 == 1) ? ev->act[2].p : ev->act[0].p;
        ev->trunkOut = x;
        ev->lastTrunkPrec = prec;
        NSG_HIP(nsg::launchHeads(x, ev->W->heads.w.p, (const float*)ev->W->heads.bias.p, (float*)ev->policy.p,
                                 ev->vfeat.p, B, ev->F, ev->headsCout, ev->vc, ev->fc1K, ev->W->heads.accScale, nsg::headPrecision(prec), s));
        const size_t partStride = (size_t)ev->batchMax * ev->vh;
        NSG_HIP(nsg::launchDense(ev->vfeat.p, ev->W->fc1.w.p, (const float*)ev->W->fc1.bias.p, (float*)ev->hidden.p,
                                 B, ev->fc1K, ev->vh, partStride, ev->W->fc1.accScale, nsg::headPrecision(prec), s));
        NSG_HIP(nsg::launchValueOut((const float*)ev->hidden.p, (const float*)ev->W->fc1.bias.p,
                                    nsg::denseSplits(ev->fc1K, nsg::headPrecision(prec)), partStride, (const float*)ev->W->fc2W.p,
                                    (const float*)ev->W->fc2B.p, (float*)ev->value.p, (float*)ev->draw.p, B, ev->vh, s));
    } else if (chains == 1) {
        int rc = enqueueChain(ev, 0, B, plan, s, true, prof ? e[1] : nullptr, prof ? e[2] : nullptr);
        if (rc) return rc;
    } else {
        // stagger: measured layer time / chains (first forward of a launch shape runs unstaggered and is timed)
        int delayUs = (stagger || ev->chainDelayUs > 0) ? ev->chainDelayUs : 0;
        bool calibrate = false;
        if (delayUs < 0) {
            if (ev->calibPending) { // the previous forward has been awaited: its events are complete
                float ms = 0.f;
                if (hipEventSynchronize(ev->calibEv[1]) == hipSuccess &&
                    hipEventElapsedTime(&ms, ev->calibEv[0], ev->calibEv[1]) == hipSuccess && ev->blocks > 0) {
                    ev->calibLayerUs = ms * 1000.f / (float)(2 * ev->blocks);
                    ev->calibKey = ev->calibPendingKey;
                }
                ev->calibPending = false;
            }
            const int wgPerChain = (per / 2) * std::max(1, ev->F / (plan.nwaves * plan.nfrag * 16));
            const int rounds = (wgPerChain + ev->prop.multiProcessorCount - 1) / ev->prop.multiProcessorCount;
            const int key = rounds * 16 + chains;
            if (key == ev->calibKey && ev->calibLayerUs > 0.f) {
                delayUs = (int)(ev->calibLayerUs / (float)chains + 0.5f);
            } else {
                delayUs = 0;
                calibrate = ev->blocks > 0;
                ev->calibPendingKey = key;
            }
        }
        NSG_HIP(hipEventRecord(ev->forkEvent, s));
        if (prof) NSG_HIP(hipEventRecord(e[1], s));
        for (int c = 0; c < chains; ++c) {
            const int off = c * per;
            const int count = std::min(per, B - off);
            if (count <= 0) break;
            hipStream_t cs = (c == 0) ? s : ev->chainStream[c - 1];
            if (c > 0) NSG_HIP(hipStreamWaitEvent(cs, ev->forkEvent, 0));
            if (c > 0 && delayUs > 0) {
                hipLaunchKernelGGL(delayKernel, dim3(1), dim3(64), 0, cs, (unsigned long long)delayUs * 100ull * c);
            }
            const bool timeIt = calibrate && c == 0;
            int rc = enqueueChain(ev, off, count, plan, cs, c == 0, timeIt ? ev->calibEv[0] : nullptr,
                                  timeIt ? ev->calibEv[1] : nullptr);
            if (rc) return rc;
            if (timeIt) ev->calibPending = true;
            if (c > 0) {
                NSG_HIP(hipEventRecord(ev->joinEvent[c - 1], cs));
                NSG_HIP(hipStreamWaitEvent(s, ev->joinEvent[c - 1], 0));
            }
        }
        if (prof) NSG_HIP(hipEventRecord(e[2], s));
    }
    if (prof) {
        NSG_HIP(hipEventRecord(e[3], s));
        ev->evUsed += 4;
        // launches bracketed by e[1]..e[2]: the 2N residual convs, or the ONE persistent launch (stem + 2N convs)
        ev->pendingTrunkLaunchesPerFwd = (chains == 1 && trunkKernel) ? 1 : 2 * ev->blocks;
    }
    return NSG_OK;
}

// Second half of every load: adopt the (possibly shared) weights and allocate this evaluator's own
// activation buffers, sized for batchMax.
int finishLoad(nsg_evaluator* ev, std::shared_ptr<NetWeights> W) {
    int rc;
    const int prec = ev->prec;
    const int es = nsg::elemSize(prec);
    ev->W = std::move(W);
    const NetWeights& N = *ev->W;
    ev->F = N.F; ev->blocks = N.blocks; ev->vc = N.vc; ev->vh = N.vh;
    ev->cpad = N.cpad; ev->headsCout = N.headsCout; ev->fc1K = N.fc1K; ev->params = N.params;
    // activation buffers: whole workgroups of up to 2 boards
    const size_t bpad = (size_t)roundUp(ev->batchMax, 2);
    if ((rc = ev->planes.alloc(bpad * 81 * ev->cpad * es, true))) return rc;
    for (int i = 0; i < 3; ++i)
        if ((rc = ev->act[i].alloc(bpad * 81 * N.F * es, true))) return rc;
    if ((rc = ev->vfeat.alloc((size_t)ev->batchMax * ev->fc1K * es, true))) return rc;
    if ((rc = ev->hidden.alloc((size_t)ev->batchMax * N.vh * 4 * nsg::denseSplits(ev->fc1K, nsg::headPrecision(prec)), true))) return rc;
    {   // persistent-trunk layer list, same buffer rotation as the per-layer path
        const int nl = 1 + 2 * N.blocks;
        std::vector<unsigned char> host((size_t)nl * nsg::trunkLayerBytes());
        void* x = ev->act[0].p; void* y = ev->act[1].p; void* z = ev->act[2].p;
        nsg::fillTrunkLayer(host.data(), 0, ev->planes.p, N.stem.w.p, (const float*)N.stem.bias.p,
                            nullptr, x, ev->cpad, N.F, 1, N.stem.accScale);
        for (int k = 0; k < N.blocks; ++k) {
            nsg::fillTrunkLayer(host.data(), 1 + 2 * k, x, N.conv1[k].w.p, (const float*)N.conv1[k].bias.p,
                                nullptr, y, N.F, N.F, 1, N.conv1[k].accScale);
            nsg::fillTrunkLayer(host.data(), 2 + 2 * k, y, N.conv2[k].w.p, (const float*)N.conv2[k].bias.p,
                                x, z, N.F, N.F, 1, N.conv2[k].accScale,
                                nsg::isMx(prec) && k == N.blocks - 1);
            void* t = x; x = z; z = t;
        }
        if ((rc = ev->trunkLayers.alloc(host.size(), false))) return rc;
        NSG_HIP(hipMemcpy(ev->trunkLayers.p, host.data(), host.size(), hipMemcpyHostToDevice));
        ev->trunkLayerCount = nl;
        const char* env = getenv("NSG_TRUNK_KERNEL");
        ev->useTrunkKernel = !env ? -1 : (env[0] == '1' ? 1 : 0);
    }
    // what the persistent launches with hand-offs share: the event behind the most recent one, the host-mapped give-up word
    if (!ev->teamDone) NSG_HIP(hipEventCreateWithFlags(&ev->teamDone, hipEventDisableTiming));
    if (!ev->teamStatusHost) {
        NSG_HIP(hipHostMalloc((void**)&ev->teamStatusHost, 64, hipHostMallocMapped));
        *ev->teamStatusHost = 0;
        NSG_HIP(hipHostGetDevicePointer((void**)&ev->teamStatusDev, ev->teamStatusHost, 0));
    }
    {   // cooperative trunk (mid batches): flags of (board, member)
        const char* env = getenv("NSG_COOP_TRUNK");
        ev->coopEnabled = (env && env[0] == '0') ? 0 : 1;
        ev->coopForced = (env && env[0] == '1') ? 1 : 0;
        if (prec == nsg::kF16m6 && (N.F == 256 || N.F == 192) && N.blocks >= 1 && ev->coopEnabled) {
            if ((rc = ev->coopFlags.alloc((size_t)ev->batchMax * 24 * sizeof(unsigned), true))) return rc;
        } else {
            ev->coopEnabled = 0;
        }
    }
    // team trunk layer list: the kF16x3 copy of the trunk (an MX evaluator keeps one for batches without an MX plan;
    // a kF16x3 evaluator's own records), same buffer rotation
    ev->teamLayerCount = 0;
    const bool mxHere = nsg::isMx(prec);
    if ((mxHere || prec == nsg::kF16x3) && N.blocks >= 1 && nsg::teamTrunkSupports(N.F, ev->cpad, 1) &&
        (!mxHere || (N.stemX3.w.p && N.conv1X3.size() == (size_t)N.blocks))) {
        const ConvLayer& stem = mxHere ? N.stemX3 : N.stem;
        const std::vector<ConvLayer>& c1 = mxHere ? N.conv1X3 : N.conv1;
        const std::vector<ConvLayer>& c2 = mxHere ? N.conv2X3 : N.conv2;
        const int nl = 1 + 2 * N.blocks;
        std::vector<nsg::TeamLayer> host((size_t)nl);
        unsigned char* x = (unsigned char*)ev->act[0].p; unsigned char* y = (unsigned char*)ev->act[1].p;
        unsigned char* z = (unsigned char*)ev->act[2].p;
        host[0] = nsg::TeamLayer{(const unsigned char*)ev->planes.p, (const nsg::team_u32x4*)stem.w.p, (const float*)stem.bias.p,
                                 nullptr, x, ev->cpad, N.F, 1, stem.accScale};
        for (int k = 0; k < N.blocks; ++k) {
            host[1 + 2 * k] = nsg::TeamLayer{x, (const nsg::team_u32x4*)c1[k].w.p, (const float*)c1[k].bias.p, nullptr, y,
                                             N.F, N.F, 1, c1[k].accScale};
            host[2 + 2 * k] = nsg::TeamLayer{y, (const nsg::team_u32x4*)c2[k].w.p, (const float*)c2[k].bias.p, x, z,
                                             N.F, N.F, 1, c2[k].accScale};
            unsigned char* t = x; x = z; z = t;
        }
        if ((rc = ev->teamLayers.alloc(host.size() * sizeof(nsg::TeamLayer), false))) return rc;
        NSG_HIP(hipMemcpy(ev->teamLayers.p, host.data(), host.size() * sizeof(nsg::TeamLayer), hipMemcpyHostToDevice));
        ev->teamBoards = ev->batchMax < nsg::kTeamMaxBoards ? ev->batchMax : nsg::kTeamMaxBoards;
        const size_t setsBytes = 2 * 4 * (size_t)ev->teamBoards * 81 * 1024;
        if ((rc = ev->teamSets.alloc(setsBytes, false))) return rc;
        NSG_HIP(hipMemset(ev->teamSets.p, 0xff, setsBytes));
        ev->teamSet = 0;
        ev->teamDirty[0] = ev->teamDirty[1] = 0;
        ev->teamLayerCount = nl;
        const char* env = getenv("NSG_TEAM_TRUNK");
        ev->teamEnabled = (env && env[0] == '0') ? 0 : 1;
    }
    if ((ev->teamLayerCount > 0 && ev->teamEnabled) || ev->coopEnabled) {
        if (!ev->teamLockHeld) {
            if (teamDeviceLock(ev->gpu)) {
                ev->teamLockHeld = true;
            } else { // another process runs persistent launches with hand-offs on this device
                ev->teamEnabled = 0;
                ev->coopEnabled = 0;
                ev->teamLockedOut = 1;
            }
        }
    }
    NSG_HIP(hipDeviceSynchronize());
    ev->calibKey = -1; // a new network: re-measure the layer time for the chain stagger
    ev->calibPending = false;
    ev->loaded = true;
    return NSG_OK;
}

int checkCompute(nsg_evaluator* ev, size_t n) {
    if (!ev) return fail(NSG_E_INVALID, "null evaluator");
    if (!ev->loaded) return fail(NSG_E_NOT_LOADED, "no weights loaded (call nsg_load first)");
    if (n == 0 || n > (size_t)ev->batchMax)
        return fail(NSG_E_INVALID, "batch size %zu outside [1, %d]", n, ev->batchMax);
    // the calling thread may never have called resetGPU for this evaluator (a pipeline's launch
    // thread, a self-play worker's second executor): every launch below must go to ev's device
    return bind(ev);
}

// What a compute call queues behind its forward: the legal-move gather and the D2H copies (trt.cc:265-271).
int enqueueOutputs(nsg_evaluator* ev) {
    const nsg_evaluator::Pending& P = ev->pending;
    if (P.kind == 1) {
        NSG_HIP(hipMemcpyAsync(P.pol, ev->policy.p, P.n * NSG_MOVE_INDEX_MAX * sizeof(float), hipMemcpyDeviceToHost, ev->stream));
    } else if (P.kind == 2) {
        if (P.total) {
            NSG_HIP(nsg::launchGatherLogits((const float*)ev->policy.p, (const uint16_t*)ev->moveIdx.p,
                                            (const uint32_t*)ev->moveOff.p, (float*)ev->gathered.p, (int)P.n, P.softmax, ev->stream));
            NSG_HIP(hipMemcpyAsync(P.vals, ev->gathered.p, P.total * sizeof(float), hipMemcpyDeviceToHost, ev->stream));
        }
    } else {
        return NSG_OK;
    }
    NSG_HIP(hipMemcpyAsync(P.win, ev->value.p, P.n * sizeof(float), hipMemcpyDeviceToHost, ev->stream));
    NSG_HIP(hipMemcpyAsync(P.draw, ev->draw.p, P.n * sizeof(float), hipMemcpyDeviceToHost, ev->stream));
    return NSG_OK;
}

// Called behind every synchronisation of ev->stream that may follow a forward.  A team launch whose members waited
// for each other in vain (a device partition smaller than the grid, another process's persistent kernel on the CUs)
// has raised the host-mapped give-up word and unwound; what it left in the output buffers is undefined.  The batch
// is still in ev->input (and the gather's indices in ev->moveIdx): run it again on the per-layer kernels, issue the
// call's copies again, wait.  The team path stays off for this evaluator from here on; nsg_get_team_stats counts.
int teamRecover(nsg_evaluator* ev) {
    if (!ev->teamStatusHost || *ev->teamStatusHost == 0) return NSG_OK;
    *ev->teamStatusHost = 0;
    if (ev->lastPersistent == 2) ev->coopEnabled = 0; // the cooperative trunk gave up: the per-layer kernels of the same plan
    else ev->teamEnabled = 0;
    ++ev->teamFallbacks;
    releaseTeamToken(ev);
    if (ev->pending.kind == 0 || ev->pending.n == 0 || !ev->teamLast) return NSG_OK;
    --ev->statBatches; // (the same batch, not a new one)
    ev->statPositions -= ev->pending.n;
    int rc = enqueueForward(ev, ev->pending.n);
    if (rc) return rc;
    if ((rc = enqueueOutputs(ev))) return rc;
    NSG_HIP(hipStreamSynchronize(ev->stream));
    return NSG_OK;
}

int syncAndRecover(nsg_evaluator* ev) {
    NSG_HIP(hipStreamSynchronize(ev->stream));
    return teamRecover(ev);
}

} // namespace

extern "C" {

const char* nsg_last_error(void) { return gLastError.c_str(); }
const char* nsg_version(void) { return "nsg 0.1 (gfx950)"; }

// The tuning variables are read when an evaluator is created; a value outside a variable's domain is an
// error there, not a silent "automatic" (a typo must not turn an A/B run into A/A).
static int checkTuningEnv() {
    struct Var { const char* name; long lo, hi; const char* what; };
    static const Var vars[] = {
        {"NSG_CONV_NB", 1, 2, "boards per workgroup"}, {"NSG_CONV_NWAVES", 1, 4, "waves per workgroup"},
        {"NSG_CONV_NFRAG", 1, 4, "fragments per wave (1, 2 or 4)"}, {"NSG_CONV_MSPLIT", 1, 2, "row split"},
        {"NSG_CHAINS", 1, nsg_evaluator::kMaxChains, "half-batch chains"}, {"NSG_TRUNK_KERNEL", 0, 1, "persistent trunk"},
        {"NSG_CHAIN_DELAY_US", -1, 1000000, "chain stagger"}, {"NSG_CHAIN_MIN_BATCH", 2, 65535, "smallest chained batch"},
        {"NSG_KSPLIT4_MAX_BATCH", 0, 65535, "largest batch of the four-way K split"},
        {"NSG_ROWSPLIT8_MAX_BATCH", 0, 65535, "largest batch of the four-way K split with two row groups"},
        {"NSG_SPLIT_BATCH", 0, 1, "full part + remainder batches"},
        {"NSG_SPLIT_BATCH_MAX", 0, 64, "largest batch (quarters of the CU count) that starts with a full chip of two-board tiles"}, {"NSG_ROCTX", 0, 1, "profiler markers"},
        {"NSG_SHARED_FORCE_COPY", 0, 1, "nsg_load_shared copies even on one device"},
        {"NSG_SLAB_SPLIT", 0, 1, "slab-split two-board tiles at mid batches"},
        {"NSG_KSPLIT3", 0, 1, "three-way K split of 192-channel nets at small and mid batches"},
        {"NSG_TEAM_TRUNK", 0, 1, "team trunk for the smallest batches"},
        {"NSG_COOP_TRUNK", 0, 1, "cooperative trunk for the two-way K split of the mid batches"},
        {"NSG_TEAM_MAX_BATCH", 0, 16, "largest batch that runs the team trunk"},
        {"NSG_TEAM_FAULT_LAUNCHES", 0, 1000000, "test hook: team launches made one workgroup short"},
        {"NSG_COOP_FAULT_XCC_LAUNCHES", 0, 1000000, "test hook: cooperative launches whose members claim different XCDs"},
        {"NSG_COOP_FLAG_BASE", 0, (1 << 24) - 1, "test hook: first flag value of the evaluator's first cooperative launch"},
        {"NSG_HEADS_SMALL_BOARDS", 0, 65535, "largest batch whose heads run four one-fragment waves per workgroup (read once per process)"},
        {"NSG_TEAM_MEMBERS", 16, 96, "most workgroups per board of the team trunk on a 256-channel net: 16, 32, 48 or 96 "
                                     "(= 1, 2, 3 or 6 row groups; a 192-channel net runs 12 per row group)"}};
    for (const Var& v : vars) {
        const char* e = getenv(v.name);
        if (!e) continue;
        char* end = nullptr;
        const long x = strtol(e, &end, 10);
        const bool bad = end == e || *end != 0 || x < v.lo || x > v.hi || (!strcmp(v.name, "NSG_CONV_NFRAG") && x == 3) ||
                         (!strcmp(v.name, "NSG_TEAM_MEMBERS") && x != 16 && x != 32 && x != 48 && x != 96);
        if (bad) return fail(NSG_E_INVALID, "%s=%s: expected an integer in [%ld, %ld] (%s)", v.name, e, v.lo, v.hi, v.what);
    }
    return NSG_OK;
}

// NSG_PRECISION: the arithmetic an evaluator starts with when the caller never calls
// nsg_set_precision -- the engine's executor ladders construct, load and run, so
// this is how an unmodified ladder reaches the benchmarked arithmetic.  A number 0..5 or a name.
static int precisionFromEnv(int* prec) {
    const char* e = getenv("NSG_PRECISION");
    if (!e || !*e) return NSG_OK;
    static const char* const names[] = {"fp32", "fp16", "bf16", "f16x3", "f16m8", "f16m6"};
    for (int i = 0; i <= NSG_PRECISION_F16M6; ++i) {
        const char digit[2] = {(char)('0' + i), 0};
        if (!strcasecmp(e, names[i]) || !strcmp(e, digit)) {
            *prec = i;
            return NSG_OK;
        }
    }
    return fail(NSG_E_INVALID, "NSG_PRECISION=%s: expected 0..5 or one of fp32 fp16 bf16 f16x3 f16m8 f16m6", e);
}

int nsg_create(int gpu_id, int batch_size_max, int num_channels, nsg_evaluator** out) {
    if (!out) return fail(NSG_E_INVALID, "null out pointer");
    *out = nullptr;
    if (int trc = checkTuningEnv()) return trc;
    int envPrec = NSG_PRECISION_FP32;
    if (int prc = precisionFromEnv(&envPrec)) return prc;
    if (batch_size_max <= 0 || batch_size_max > 65535 || num_channels <= 0 || num_channels > 1024)
        return fail(NSG_E_INVALID, "bad batch_size_max/num_channels");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return fail(NSG_E_HIP, "no HIP device available (%s): this library has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (gpu_id < 0 || gpu_id >= count) return fail(NSG_E_INVALID, "gpu_id %d out of range", gpu_id);
    std::unique_ptr<nsg_evaluator> ev(new nsg_evaluator());
    ev->gpu = gpu_id;
    ev->batchMax = batch_size_max;
    ev->numChannels = num_channels;
    ev->prec = envPrec;
    NSG_HIP(hipSetDevice(gpu_id));
    NSG_HIP(hipGetDeviceProperties(&ev->prop, gpu_id));
    int rc;
    // trt.cc:57-77: device buffers sized for BatchSizeMax, zero-initialised
    if ((rc = ev->input.alloc((size_t)batch_size_max * num_channels * NSG_BITBOARD_BYTES, true))) return rc;
    if ((rc = ev->policy.alloc((size_t)batch_size_max * NSG_MOVE_INDEX_MAX * 4, true))) return rc;
    if ((rc = ev->value.alloc((size_t)batch_size_max * 4, true))) return rc;
    if ((rc = ev->draw.alloc((size_t)batch_size_max * 4, true))) return rc;
    // trt.cc:79
    NSG_HIP(hipStreamCreateWithFlags(&ev->stream, hipStreamNonBlocking));
    for (auto& cs : ev->chainStream) NSG_HIP(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    NSG_HIP(hipEventCreateWithFlags(&ev->forkEvent, hipEventDisableTiming));
    for (auto& je : ev->joinEvent) NSG_HIP(hipEventCreateWithFlags(&je, hipEventDisableTiming));
    if (const char* e2 = getenv("NSG_CHAINS")) {
        const int v = atoi(e2);
        if (v >= 1 && v <= nsg_evaluator::kMaxChains) ev->numChains = v;
    }
    ev->tuning = nsg::readConvTuning();
    for (auto& ce : ev->calibEv) NSG_HIP(hipEventCreate(&ce));
    if (const char* e2 = getenv("NSG_CHAIN_DELAY_US")) ev->chainDelayUs = std::max(-1, atoi(e2));
    if (const char* e2 = getenv("NSG_CHAIN_MIN_BATCH")) ev->chainMinBatch = std::max(2, atoi(e2));
    if (const char* e2 = getenv("NSG_TEAM_MEMBERS")) ev->teamForceRowGroups = atoi(e2) / 16; // (validated above)
    if (const char* e2 = getenv("NSG_TEAM_FAULT_LAUNCHES")) ev->teamFaultLaunches = std::max(0, atoi(e2));
    if (const char* e2 = getenv("NSG_COOP_FAULT_XCC_LAUNCHES")) ev->coopFaultXccLaunches = std::max(0, atoi(e2));
    if (const char* e2 = getenv("NSG_COOP_FLAG_BASE")) ev->coopFlagBase = (unsigned)std::max(0, atoi(e2));
    if (const char* e2 = getenv("NSG_TEAM_MAX_BATCH")) ev->teamMaxBatch = std::min(std::max(0, atoi(e2)), (int)nsg::kTeamMaxBoards);
    *out = ev.release();
    return NSG_OK;
}

int nsg_destroy(nsg_evaluator* ev) {
    if (!ev) return NSG_OK;
    (void)hipSetDevice(ev->gpu);
    if (ev->stream) {
        (void)hipStreamSynchronize(ev->stream);
    }
    if (ev->teamStatusHost) *ev->teamStatusHost = 0;
    releaseTeamToken(ev);
    if (ev->teamLockHeld) teamDeviceUnlock(ev->gpu);
#ifdef TEAM_STAMPS
    if (ev->teamLayerCount) nsg::teamTrunkDumpStamps();
#endif
    if (ev->teamStatusHost) (void)hipHostFree(ev->teamStatusHost);
    if (ev->teamDone) (void)hipEventDestroy(ev->teamDone); // (a stream that waits for it keeps what it needs)
    for (hipEvent_t e : ev->ev) (void)hipEventDestroy(e);
    for (auto cs : ev->chainStream)
        if (cs) { (void)hipStreamSynchronize(cs); (void)hipStreamDestroy(cs); }
    if (ev->forkEvent) (void)hipEventDestroy(ev->forkEvent);
    for (auto je : ev->joinEvent)
        if (je) (void)hipEventDestroy(je);
    for (auto ce : ev->calibEv)
        if (ce) (void)hipEventDestroy(ce);
    if (ev->stream) (void)hipStreamDestroy(ev->stream);
    delete ev;
    return NSG_OK;
}

int nsg_set_precision(nsg_evaluator* ev, int precision) {
    if (!ev) return fail(NSG_E_INVALID, "null evaluator");
    if (ev->loaded) return fail(NSG_E_INVALID, "precision must be chosen before nsg_load");
    if (precision < NSG_PRECISION_FP32 || precision > NSG_PRECISION_F16M6)
        return fail(NSG_E_INVALID, "unknown precision %d", precision);
    ev->prec = precision;
    return NSG_OK;
}

int nsg_convert_onnx(const void* onnx, size_t size, void* dst, size_t capacity, size_t* nsgw_size) {
    if (!onnx || !nsgw_size) return fail(NSG_E_INVALID, "null argument");
    std::vector<unsigned char> blob;
    std::string err;
    if (!nsg::onnx::convertToNsgw(onnx, size, &blob, &err)) return fail(NSG_E_FORMAT, "%s", err.c_str());
    *nsgw_size = blob.size();
    if (dst) {
        if (capacity < blob.size()) return fail(NSG_E_INVALID, "destination holds %zu bytes, %zu needed", capacity, blob.size());
        memcpy(dst, blob.data(), blob.size());
    }
    return NSG_OK;
}

int nsg_load_memory(nsg_evaluator* ev, const void* blob, size_t size) {
    if (!ev || !blob) return fail(NSG_E_INVALID, "null argument");
    if (!nsg::onnx::isNsgw(blob, size)) { // what the engine passes: an ONNX model (trt.cc:121-131)
        std::vector<unsigned char> converted;
        std::string err;
        if (!nsg::onnx::convertToNsgw(blob, size, &converted, &err))
            return fail(NSG_E_FORMAT, "Failed to parse the model: neither an NSGW v1 weight file nor an ONNX model of the "
                                      "supported topology (%s)", err.c_str());
        return nsg_load_memory(ev, converted.data(), converted.size());
    }
    int rc = bind(ev);
    if (rc) return rc;
    NetView nv;
    if ((rc = parseBlob(blob, size, &nv))) return rc;
    if (nv.cin != ev->numChannels)
        return fail(NSG_E_FORMAT, "weight file expects %d input planes, evaluator has %d", nv.cin,
                    ev->numChannels);
    // trt.cc:193-210: the policy output must have ml::MoveIndexMax elements
    if (nv.pc * NSG_NUM_SQUARES != NSG_MOVE_INDEX_MAX)
        return fail(NSG_E_FORMAT, "Unexpected PolicySize: %d (expected: %d).",
                    nv.pc * NSG_NUM_SQUARES, NSG_MOVE_INDEX_MAX);
    if (nv.F % 64 != 0) return fail(NSG_E_FORMAT, "trunk width %d is not a multiple of 64", nv.F);
    if (nv.vh % 64 != 0) return fail(NSG_E_FORMAT, "value hidden width %d is not a multiple of 64", nv.vh);

    const int prec = ev->prec;
    const int kc = nsg::inputChannelGranule(prec); // input channels are padded to whole chunks (kF16m8: chunk pairs)
    ev->loaded = false;
    auto W = std::make_shared<NetWeights>();
    W->gpu = ev->gpu; W->prec = prec;
    W->F = nv.F; W->blocks = nv.blocks; W->vc = nv.vc; W->vh = nv.vh; W->cin = nv.cin;
    W->cpad = roundUp(nv.cin, kc);
    W->headsCout = roundUp(nv.vc + nv.pc, 64);
    W->fc1K = roundUp(81 * nv.vc, nsg::chunkChannels(nsg::headPrecision(prec)));
    W->params = nv.params;

    std::vector<double> scale;
    std::vector<float> bias;
    {   // 3x3 layers: BN folding and fragment packing on worker threads, uploads on this one
        struct Job {
            std::vector<double> scale;
            std::vector<float> bias;
            const float* w;
            int cinReal, kdim, prec;
            ConvLayer* L;
            std::vector<unsigned char> host;
            float accScale = 1.f;
        };
        W->conv1.resize(nv.blocks); W->conv2.resize(nv.blocks);
        if (nsg::isMx(prec)) { W->conv1X3.resize(nv.blocks); W->conv2X3.resize(nv.blocks); }
        std::vector<Job> jobs;
        auto add = [&](const float* w, const float* bn, int cinReal, int kdim, ConvLayer* L, ConvLayer* Lx3) {
            Job j;
            foldBn(bn, nv.F, nv.eps, &j.scale, &j.bias);
            j.w = w; j.cinReal = cinReal; j.kdim = kdim; j.prec = prec; j.L = L;
            jobs.push_back(j);
            if (nsg::isMx(prec)) { // the f16x3 copy of the trunk (small batches)
                j.prec = nsg::kF16x3; j.L = Lx3;
                jobs.push_back(std::move(j));
            }
        };
        add(nv.stemW, nv.stemBn, nv.cin, W->cpad, &W->stem, &W->stemX3);
        for (int k = 0; k < nv.blocks; ++k) {
            add(nv.w1[k], nv.bn1[k], nv.F, nv.F, &W->conv1[k], nsg::isMx(prec) ? &W->conv1X3[k] : nullptr);
            add(nv.w2[k], nv.bn2[k], nv.F, nv.F, &W->conv2[k], nsg::isMx(prec) ? &W->conv2X3[k] : nullptr);
        }
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t i; (i = next.fetch_add(1)) < jobs.size();) {
                Job& j = jobs[i];
                Conv3Ctx c{j.w, j.scale.data(), j.cinReal};
                packLayer(conv3Get, &c, 9, j.cinReal, j.kdim, nv.F, nv.F, j.prec, &j.host, &j.accScale);
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nthreads = std::min<size_t>(jobs.size(), std::max(1u, std::min(hw ? hw : 1u, 8u)));
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nthreads; ++t) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
        for (Job& j : jobs) {
            if ((rc = commitLayer(j.host, j.accScale, j.kdim, nv.F, j.bias, j.L))) return rc;
            std::vector<unsigned char>().swap(j.host);
        }
    }
    // heads: [value conv (BN folded) | policy conv | zero pad]
    {
        foldBn(nv.valBn, nv.vc, nv.eps, &scale, &bias);
        std::vector<float> hb(W->headsCout, 0.f);
        for (int i = 0; i < nv.vc; ++i) hb[i] = bias[i];
        for (int i = 0; i < nv.pc; ++i) hb[nv.vc + i] = nv.polB[i];
        HeadsCtx c{nv.valW, scale.data(), nv.polW, nv.F, nv.vc, nv.pc};
        if ((rc = uploadLayer(headsGet, &c, 1, nv.F, nv.F, W->headsCout, nv.vc + nv.pc, nsg::headPrecision(prec), hb, &W->heads))) return rc;
    }
    {
        std::vector<float> b1(nv.fc1B, nv.fc1B + nv.vh);
        Fc1Ctx c{nv.fc1W, nv.vc};
        if ((rc = uploadLayer(fc1Get, &c, 1, 81 * nv.vc, W->fc1K, nv.vh, nv.vh, nsg::headPrecision(prec), b1, &W->fc1))) return rc;
    }
    {   // activation bound estimate (see pushConvMoment)
        std::vector<double> sc;
        std::vector<float> bi;
        double worst = 0.0, mean = 0.0, peak2 = 0.0;
        // feature planes are 0/1 (a handful of scalar planes in [0,1]): second moment <= the plane density, ~0.2
        foldBn(nv.stemBn, nv.F, nv.eps, &sc, &bi);
        pushConvMoment(nv.stemW, sc.data(), bi.data(), nv.F, nv.cin, 9, 0.2, &worst, &mean);
        peak2 = worst;
        double x2 = 0.5 * mean; // after the ReLU
        for (int k = 0; k < nv.blocks; ++k) {
            foldBn(nv.bn1[k], nv.F, nv.eps, &sc, &bi);
            pushConvMoment(nv.w1[k], sc.data(), bi.data(), nv.F, nv.F, 9, x2, &worst, &mean);
            peak2 = std::max(peak2, worst);
            const double y2 = 0.5 * mean;
            foldBn(nv.bn2[k], nv.F, nv.eps, &sc, &bi);
            pushConvMoment(nv.w2[k], sc.data(), bi.data(), nv.F, nv.F, 9, y2, &worst, &mean);
            peak2 = std::max(peak2, worst + x2); // the residual add
            x2 = 0.5 * (mean + x2);
        }
        W->actBound = 6.0 * std::sqrt(peak2);
        const char* off = getenv("NSG_M8_GUARD");
        W->outsideM8Window = prec == nsg::kF16m8 && W->actBound > 224.0 && !(off && off[0] == '0');
    }
    if ((rc = W->fc2W.alloc((size_t)2 * nv.vh * 4, false))) return rc;
    if ((rc = W->fc2B.alloc(8, false))) return rc;
    NSG_HIP(hipMemcpy(W->fc2W.p, nv.fc2W, (size_t)2 * nv.vh * 4, hipMemcpyHostToDevice));
    NSG_HIP(hipMemcpy(W->fc2B.p, nv.fc2B, 8, hipMemcpyHostToDevice));
    return finishLoad(ev, std::move(W));
}

// Loads `ev` with the network `src` already holds, without touching the model file again: the
// self-play driver's G x T executors (selfplay/main.cc:189-195, mcts/manager.cc:168-179 -- where
// every executor re-reads and re-builds the model) share ONE upload per device and one peer copy
// (hipMemcpyPeer over xGMI) per further device.
int nsg_load_shared(nsg_evaluator* ev, nsg_evaluator* src) {
    if (!ev || !src) return fail(NSG_E_INVALID, "null argument");
    if (!src->loaded || !src->W) return fail(NSG_E_NOT_LOADED, "the source evaluator has no weights loaded");
    if (ev == src) return NSG_OK;
    if (ev->prec != src->W->prec) return fail(NSG_E_INVALID, "precision differs from the source evaluator's");
    if (ev->numChannels != src->W->cin)
        return fail(NSG_E_FORMAT, "network expects %d input planes, evaluator has %d", src->W->cin, ev->numChannels);
    int rc = bind(ev);
    if (rc) return rc;
    ev->loaded = false;
    // NSG_SHARED_FORCE_COPY=1: take the other-device branch (own allocations + hipMemcpyPeer) for an evaluator of
    // the SAME device too, so the branch a multi-GPU process runs can be exercised on a one-GPU machine.
    static const bool forceCopy = [] { const char* e = getenv("NSG_SHARED_FORCE_COPY"); return e && atoi(e) == 1; }();
    if (ev->gpu == src->gpu && !forceCopy) return finishLoad(ev, src->W);
    const NetWeights& S = *src->W;
    auto W = std::make_shared<NetWeights>();
    W->gpu = ev->gpu; W->prec = S.prec;
    W->F = S.F; W->blocks = S.blocks; W->vc = S.vc; W->vh = S.vh; W->cin = S.cin; W->cpad = S.cpad;
    W->headsCout = S.headsCout; W->fc1K = S.fc1K; W->params = S.params;
    W->actBound = S.actBound; W->outsideM8Window = S.outsideM8Window;
    auto copyBuf = [&](const DevBuf& s, DevBuf* d) -> int {
        if (!s.p) return NSG_OK;
        int r = d->alloc(s.bytes, false);
        if (r) return r;
        NSG_HIP(hipMemcpyPeer(d->p, ev->gpu, s.p, S.gpu, s.bytes));
        return NSG_OK;
    };
    auto copyLayer = [&](const ConvLayer& s, ConvLayer* d) -> int {
        d->cin = s.cin; d->accScale = s.accScale;
        int r = copyBuf(s.w, &d->w);
        return r ? r : copyBuf(s.bias, &d->bias);
    };
    auto copyLayers = [&](const std::vector<ConvLayer>& s, std::vector<ConvLayer>* d) -> int {
        d->resize(s.size());
        for (size_t i = 0; i < s.size(); ++i) { int r = copyLayer(s[i], &(*d)[i]); if (r) return r; }
        return NSG_OK;
    };
    if ((rc = copyLayer(S.stem, &W->stem)) || (rc = copyLayer(S.stemX3, &W->stemX3)) ||
        (rc = copyLayers(S.conv1, &W->conv1)) || (rc = copyLayers(S.conv2, &W->conv2)) ||
        (rc = copyLayers(S.conv1X3, &W->conv1X3)) || (rc = copyLayers(S.conv2X3, &W->conv2X3)) ||
        (rc = copyLayer(S.heads, &W->heads)) || (rc = copyLayer(S.fc1, &W->fc1)) ||
        (rc = copyBuf(S.fc2W, &W->fc2W)) || (rc = copyBuf(S.fc2B, &W->fc2B)))
        return rc;
    NSG_HIP(hipDeviceSynchronize());
    return finishLoad(ev, std::move(W));
}

int nsg_load(nsg_evaluator* ev, const char* path) {
    if (!ev || !path) return fail(NSG_E_INVALID, "null argument");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(NSG_E_IO, "Could not open the file: %s", path); // trt.cc:34-36
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz <= 0) {
        fclose(f);
        return fail(NSG_E_IO, "empty weight file: %s", path);
    }
    std::vector<unsigned char> blob((size_t)sz);
    const size_t got = fread(blob.data(), 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) return fail(NSG_E_IO, "short read on %s", path);
    return nsg_load_memory(ev, blob.data(), blob.size());
}

int nsg_load_device_blob(nsg_evaluator* ev, const void* device_blob, size_t size) {
    if (!ev || !device_blob) return fail(NSG_E_INVALID, "null argument");
    int rc = bind(ev);
    if (rc) return rc;
    std::vector<unsigned char> blob(size);
    NSG_HIP(hipMemcpy(blob.data(), device_blob, size, hipMemcpyDeviceToHost));
    return nsg_load_memory(ev, blob.data(), size);
}

int nsg_compute_nonblocking(nsg_evaluator* ev, const void* features, size_t batch_size,
                            float* dst_policy, float* dst_win_rate, float* dst_draw_rate) {
    int rc = checkCompute(ev, batch_size);
    if (rc) return rc;
    if (!features || !dst_policy || !dst_win_rate || !dst_draw_rate)
        return fail(NSG_E_INVALID, "null buffer");
    // trt.cc:237-238: exactly one batch in flight per executor
    if (hipStreamQuery(ev->stream) == hipErrorNotReady)
        return fail(NSG_E_BUSY, "computeNonBlocking called while a batch is in flight");
    // trt.cc:240-242
    {
        Range r("nsg.h2d");
        NSG_HIP(hipMemcpyAsync(ev->input.p, features,
                               batch_size * ev->numChannels * NSG_BITBOARD_BYTES,
                               hipMemcpyHostToDevice, ev->stream));
    }
    ev->pending = nsg_evaluator::Pending{1, batch_size, 0, 0, dst_policy, dst_win_rate, dst_draw_rate, nullptr};
    if ((rc = enqueueForward(ev, batch_size))) return rc;
    Range d2h("nsg.d2h");
    return enqueueOutputs(ev); // trt.cc:265-271
}

int nsg_compute_gather_nonblocking(nsg_evaluator* ev, const void* features, size_t batch_size,
                                   const uint16_t* move_indices, const uint32_t* move_offsets,
                                   int softmax, float* dst_values, float* dst_win_rate,
                                   float* dst_draw_rate) {
    int rc = checkCompute(ev, batch_size);
    if (rc) return rc;
    if (!features || !move_indices || !move_offsets || !dst_values || !dst_win_rate || !dst_draw_rate)
        return fail(NSG_E_INVALID, "null buffer");
    if (move_offsets[0] != 0) return fail(NSG_E_INVALID, "move_offsets[0] must be 0");
    for (size_t b = 0; b < batch_size; ++b)
        if (move_offsets[b + 1] < move_offsets[b] || move_offsets[b + 1] - move_offsets[b] > NSG_MAX_LEGAL_MOVES)
            return fail(NSG_E_INVALID, "move_offsets must be non-decreasing with at most %d moves per position",
                        NSG_MAX_LEGAL_MOVES);
    const size_t total = move_offsets[batch_size];
    if (hipStreamQuery(ev->stream) == hipErrorNotReady)
        return fail(NSG_E_BUSY, "compute called while a batch is in flight");
    if (!ev->moveIdx.p) {
        const size_t cap = (size_t)ev->batchMax * NSG_MAX_LEGAL_MOVES;
        if ((rc = ev->moveIdx.alloc(cap * sizeof(uint16_t), false))) return rc;
        if ((rc = ev->moveOff.alloc(((size_t)ev->batchMax + 1) * sizeof(uint32_t), false))) return rc;
        if ((rc = ev->gathered.alloc(cap * sizeof(float), false))) return rc;
    }
    NSG_HIP(hipMemcpyAsync(ev->input.p, features, batch_size * ev->numChannels * NSG_BITBOARD_BYTES,
                           hipMemcpyHostToDevice, ev->stream));
    NSG_HIP(hipMemcpyAsync(ev->moveOff.p, move_offsets, (batch_size + 1) * sizeof(uint32_t),
                           hipMemcpyHostToDevice, ev->stream));
    if (total)
        NSG_HIP(hipMemcpyAsync(ev->moveIdx.p, move_indices, total * sizeof(uint16_t), hipMemcpyHostToDevice,
                               ev->stream));
    ev->pending = nsg_evaluator::Pending{2, batch_size, total, softmax ? 1 : 0, nullptr, dst_win_rate, dst_draw_rate, dst_values};
    if ((rc = enqueueForward(ev, batch_size))) return rc;
    return enqueueOutputs(ev);
}

int nsg_compute_gather_blocking(nsg_evaluator* ev, const void* features, size_t batch_size,
                                const uint16_t* move_indices, const uint32_t* move_offsets,
                                int softmax, float* dst_values, float* dst_win_rate,
                                float* dst_draw_rate) {
    int rc = nsg_compute_gather_nonblocking(ev, features, batch_size, move_indices, move_offsets, softmax,
                                            dst_values, dst_win_rate, dst_draw_rate);
    if (rc) return rc;
    return nsg_await(ev);
}

int nsg_await(nsg_evaluator* ev) {
    if (!ev) return fail(NSG_E_INVALID, "null evaluator");
    int rc = bind(ev);
    if (rc) return rc;
    // trt.cc:281-283; a team launch that gave up is re-run on the per-layer kernels before this returns (teamRecover):
    // the caller sees a slow batch, not a lost one
    return syncAndRecover(ev);
}

int nsg_compute_blocking(nsg_evaluator* ev, const void* features, size_t batch_size,
                         float* dst_policy, float* dst_win_rate, float* dst_draw_rate) {
    int rc = nsg_compute_nonblocking(ev, features, batch_size, dst_policy, dst_win_rate,
                                     dst_draw_rate); // trt.cc:274-279
    if (rc) return rc;
    return nsg_await(ev);
}

int nsg_is_computing(nsg_evaluator* ev) {
    if (!ev) return 0;
    return hipStreamQuery(ev->stream) == hipErrorNotReady ? 1 : 0; // trt.cc:285-287
}

int nsg_reset_gpu(nsg_evaluator* ev) {
    if (!ev) return fail(NSG_E_INVALID, "null evaluator");
    return bind(ev); // trt.cc:289-291
}

int nsg_extract_bits(float* dst, const uint64_t* src, int batch_size, int num_channels,
                     int channels_first, void* hip_stream) {
    if (!dst || !src || batch_size <= 0 || num_channels <= 0)
        return fail(NSG_E_INVALID, "bad extract_bits argument"); // extractbit.cu:78 assert
    if (!channels_first && num_channels > 1024)
        return fail(NSG_E_INVALID, "NumChannels > 1024"); // extractbit.cu:91
    hipStream_t s = (hipStream_t)hip_stream;
    if (channels_first) {
        NSG_HIP(nsg::launchExtractBitsNCHW(dst, src, batch_size, num_channels, s));
    } else {
        NSG_HIP(nsg::launchExtractBitsNHWC(dst, src, batch_size, num_channels, s));
    }
    return NSG_OK;
}

int nsg_host_register(void* ptr, size_t bytes) {
    if (!ptr || bytes == 0) return fail(NSG_E_INVALID, "bad host_register argument");
    NSG_HIP(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return NSG_OK;
}

int nsg_host_unregister(void* ptr) {
    if (!ptr) return fail(NSG_E_INVALID, "null pointer");
    NSG_HIP(hipHostUnregister(ptr));
    return NSG_OK;
}

int nsg_upload_features(nsg_evaluator* ev, const void* features, size_t batch_size) {
    if (!ev || !features || batch_size == 0 || batch_size > (size_t)ev->batchMax)
        return fail(NSG_E_INVALID, "bad upload_features argument");
    int rc = bind(ev);
    if (rc) return rc;
    if ((rc = syncAndRecover(ev))) return rc; // (before the batch a forward in flight still reads is overwritten)
    ev->pending = nsg_evaluator::Pending{};
    NSG_HIP(hipMemcpyAsync(ev->input.p, features, batch_size * ev->numChannels * NSG_BITBOARD_BYTES,
                           hipMemcpyHostToDevice, ev->stream));
    NSG_HIP(hipStreamSynchronize(ev->stream));
    return NSG_OK;
}

int nsg_forward_resident(nsg_evaluator* ev, size_t batch_size) {
    int rc = checkCompute(ev, batch_size);
    if (rc) return rc;
    ev->pending = nsg_evaluator::Pending{3, batch_size, 0, 0, nullptr, nullptr, nullptr, nullptr};
    return enqueueForward(ev, batch_size);
}

int nsg_download_outputs(nsg_evaluator* ev, size_t batch_size, float* dst_policy,
                         float* dst_win_rate, float* dst_draw_rate) {
    int rc = checkCompute(ev, batch_size);
    if (rc) return rc;
    if ((rc = syncAndRecover(ev))) return rc;
    NSG_HIP(hipMemcpyAsync(dst_policy, ev->policy.p, batch_size * NSG_MOVE_INDEX_MAX * sizeof(float),
                           hipMemcpyDeviceToHost, ev->stream));
    NSG_HIP(hipMemcpyAsync(dst_win_rate, ev->value.p, batch_size * sizeof(float),
                           hipMemcpyDeviceToHost, ev->stream));
    NSG_HIP(hipMemcpyAsync(dst_draw_rate, ev->draw.p, batch_size * sizeof(float),
                           hipMemcpyDeviceToHost, ev->stream));
    NSG_HIP(hipStreamSynchronize(ev->stream));
    return NSG_OK;
}

int nsg_download_trunk(nsg_evaluator* ev, size_t batch_size, float* dst) {
    int rc = checkCompute(ev, batch_size);
    if (rc) return rc;
    if (!ev->trunkOut) return fail(NSG_E_INVALID, "no forward has run yet");
    if ((rc = syncAndRecover(ev))) return rc;
    const size_t bytes = batch_size * ev->F * 81 * sizeof(float);
    if (ev->scratch.bytes < bytes && (rc = ev->scratch.alloc(bytes, false))) return rc;
    NSG_HIP(nsg::launchActToNCHW(ev->trunkOut, (float*)ev->scratch.p, (int)batch_size, ev->F,
                                 nsg::headPrecision(ev->prec), ev->stream));
    NSG_HIP(hipMemcpyAsync(dst, ev->scratch.p, bytes, hipMemcpyDeviceToHost, ev->stream));
    NSG_HIP(hipStreamSynchronize(ev->stream));
    return NSG_OK;
}

int nsg_download_planes_raw(nsg_evaluator* ev, size_t batch_size, void* dst, size_t capacity, size_t* row_bytes) {
    int rc = checkCompute(ev, batch_size);
    if (rc) return rc;
    if (!ev->trunkOut) return fail(NSG_E_INVALID, "no forward has run yet");
    if ((rc = syncAndRecover(ev))) return rc;
    if (ev->teamLast && ev->lastPersistent == 1)
        return fail(NSG_E_INVALID, "the last forward ran the team trunk, which decodes the bitboards inside its first layer: "
                                   "there is no plane buffer for it (NSG_TEAM_TRUNK=0 keeps the per-layer kernels)");
    if (!dst || !row_bytes) return fail(NSG_E_INVALID, "null argument");
    const size_t es = nsg::elemSize(ev->lastTrunkPrec);
    const size_t rb = (size_t)ev->cpad * es;
    const size_t bytes = batch_size * 81 * rb;
    if (capacity < bytes) return fail(NSG_E_INVALID, "destination holds %zu bytes, %zu needed", capacity, bytes);
    NSG_HIP(hipMemcpyAsync(dst, ev->planes.p, bytes, hipMemcpyDeviceToHost, ev->stream));
    NSG_HIP(hipStreamSynchronize(ev->stream));
    *row_bytes = rb;
    return NSG_OK;
}

int nsg_time_planes(nsg_evaluator* ev, size_t batch_size, int iterations, float* avg_ms, double* bytes_per_launch) {
    int rc = checkCompute(ev, batch_size);
    if (rc) return rc;
    if (iterations < 1 || !avg_ms) return fail(NSG_E_INVALID, "bad time_planes argument");
    if ((rc = syncAndRecover(ev))) return rc;
    const int prec = ev->prec;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    NSG_HIP(hipEventCreate(&e0));
    NSG_HIP(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i)
        NSG_HIP(nsg::launchExtractBitsAct(ev->planes.p, (const uint64_t*)ev->input.p, (int)batch_size, ev->numChannels, ev->cpad, prec, ev->stream));
    NSG_HIP(hipEventRecord(e0, ev->stream));
    for (int i = 0; i < iterations; ++i)
        NSG_HIP(nsg::launchExtractBitsAct(ev->planes.p, (const uint64_t*)ev->input.p, (int)batch_size, ev->numChannels, ev->cpad, prec, ev->stream));
    NSG_HIP(hipEventRecord(e1, ev->stream));
    NSG_HIP(hipEventSynchronize(e1));
    float ms = 0.f;
    NSG_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    *avg_ms = ms / (float)iterations;
    // algorithmic bytes: the bitboards read + the trunk-input rows written ([81][cpad] elements per position)
    if (bytes_per_launch)
        *bytes_per_launch = (double)batch_size * ((double)ev->numChannels * NSG_BITBOARD_BYTES + 81.0 * ev->cpad * nsg::elemSize(prec));
    return NSG_OK;
}

int nsg_profile_enable(nsg_evaluator* ev, int enable) {
    if (!ev) return fail(NSG_E_INVALID, "null evaluator");
    int rc = bind(ev);
    if (rc) return rc;
    if (enable && ev->ev.empty()) {
        ev->ev.resize(kMaxEventPairs * 4);
        for (auto& e : ev->ev) NSG_HIP(hipEventCreate(&e));
    }
    if (!enable && (rc = drainProfile(ev))) return rc;
    ev->profile = enable != 0;
    return NSG_OK;
}

int nsg_profile_read(nsg_evaluator* ev, double* trunk_ms_total, uint64_t* trunk_launches,
                     double* forward_ms_total, uint64_t* forwards) {
    if (!ev) return fail(NSG_E_INVALID, "null evaluator");
    int rc = drainProfile(ev);
    if (rc) return rc;
    if (trunk_ms_total) *trunk_ms_total = ev->trunkMs;
    if (trunk_launches) *trunk_launches = ev->trunkLaunches;
    if (forward_ms_total) *forward_ms_total = ev->fwdMs;
    if (forwards) *forwards = ev->forwards;
    ev->trunkMs = ev->fwdMs = 0;
    ev->trunkLaunches = ev->forwards = 0;
    return NSG_OK;
}

int nsg_get_stats(nsg_evaluator* ev, uint64_t* batches, uint64_t* positions) {
    if (!ev) return fail(NSG_E_INVALID, "null evaluator");
    if (batches) *batches = ev->statBatches;
    if (positions) *positions = ev->statPositions;
    return NSG_OK;
}

int nsg_get_team_stats(nsg_evaluator* ev, int* enabled, int* members_last, uint64_t* fallbacks) {
    if (!ev) return fail(NSG_E_INVALID, "null evaluator");
    if (enabled) *enabled = (ev->teamLayerCount > 0 && ev->teamEnabled) ? 1 : (ev->teamLockedOut ? -1 : 0);
    if (members_last) *members_last = ev->teamLastMembers;
    if (fallbacks) *fallbacks = ev->teamFallbacks;
    return NSG_OK;
}

int nsg_get_last_launch_kind(nsg_evaluator* ev, int* kind, int* coop_enabled) {
    if (!ev) return fail(NSG_E_INVALID, "null evaluator");
    if (kind) *kind = ev->lastPersistent;
    if (coop_enabled) *coop_enabled = ev->coopEnabled ? 1 : (ev->teamLockedOut ? -1 : 0);
    return NSG_OK;
}

int nsg_get_last_plan(nsg_evaluator* ev, int* nb, int* nfrag, int* nwaves, int* chains) {
    if (!ev) return fail(NSG_E_INVALID, "null evaluator");
    if (nb) *nb = ev->lastPlan.nb;
    if (nfrag) *nfrag = ev->lastPlan.nfrag;
    if (nwaves) *nwaves = ev->lastPlan.nwaves;
    if (chains) *chains = ev->lastChains;
    return NSG_OK;
}

int nsg_get_last_split(nsg_evaluator* ev, int* row_split, int* k_split) {
    if (!ev) return fail(NSG_E_INVALID, "null evaluator");
    if (row_split) *row_split = ev->lastPlan.nb ? ev->lastPlan.msplit : 0;
    if (k_split) *k_split = ev->lastPlan.nb ? ev->lastPlan.ksplit : 0;
    return NSG_OK;
}

int nsg_get_last_slab_split(nsg_evaluator* ev, int* slab_split) {
    if (!ev || !slab_split) return fail(NSG_E_INVALID, "null argument");
    *slab_split = ev->lastPlan.nb ? ev->lastPlan.sslab : 0;
    return NSG_OK;
}

int nsg_get_last_trunk_precision(nsg_evaluator* ev, int* precision) {
    if (!ev || !precision) return fail(NSG_E_INVALID, "null argument");
    *precision = ev->lastTrunkPrec;
    return NSG_OK;
}

int nsg_get_info(nsg_evaluator* ev, nsg_info* info) {
    if (!ev || !info) return fail(NSG_E_INVALID, "null argument");
    memset(info, 0, sizeof(*info));
    info->gpu_id = ev->gpu;
    info->batch_size_max = ev->batchMax;
    info->num_channels = ev->numChannels;
    info->channels = ev->F;
    info->blocks = ev->blocks;
    info->value_channels = ev->vc;
    info->value_hidden = ev->vh;
    info->precision = ev->prec;
    info->loaded = ev->loaded ? 1 : 0;
    info->compute_units = ev->prop.multiProcessorCount;
    info->clock_khz = ev->prop.clockRate;
    info->param_count = ev->params;
    const double F = ev->F, N = ev->blocks, C = ev->numChannels;
    // stem + trunk + 1x1 policy (value/draw heads excluded)
    info->flops_per_position = 2.0 * 81 * 9 * C * F + N * 2 * (2.0 * 81 * 9 * F * F) + 2.0 * 81 * 27 * F;
    info->trunk_conv_flops_per_position = 2.0 * 81 * 9 * F * F;
    info->activation_bound_estimate = ev->W ? ev->W->actBound : 0.0;
    info->f16m8_window_fallback = (ev->W && ev->W->outsideM8Window) ? 1 : 0;
    // (boxes without the amdgpu.ids table report an empty marketing name: fall back to the ISA name)
    snprintf(info->device_name, sizeof(info->device_name), "%s", ev->prop.name[0] ? ev->prop.name : ev->prop.gcnArchName);
    return NSG_OK;
}

#ifdef NSG_DIAG_STAMPS
// Diagnostic library only (make diag): cycle stamps of every trunk-conv workgroup.
// layout: [layer][4096 workgroups][8 u64] = entry, prologue done, main loop done,
// epilogue done (s_memtime), -, -, s_memrealtime at epilogue done / at entry.
int nsg_debug_stamps_enable(nsg_evaluator* ev) {
    if (!ev || !ev->loaded) return fail(NSG_E_INVALID, "load first");
    if (int rc = ev->stamps.alloc((size_t)2 * ev->blocks * 4096 * 8 * 8, true)) return rc;
    // the persistent launches (trunk kernel, cooperative trunk) take their layers from the device list: layer l > 0 stamps
    // where the per-layer launch of that convolution would
    for (int l = 1; l < ev->trunkLayerCount; ++l) {
        unsigned long long* st = (unsigned long long*)ev->stamps.p + (size_t)(l - 1) * 4096 * 8;
        NSG_HIP(hipMemcpy((unsigned char*)ev->trunkLayers.p + (size_t)l * nsg::trunkLayerBytes() + nsg::trunkLayerStampsOffset(), &st,
                          sizeof(st), hipMemcpyHostToDevice));
    }
    return NSG_OK;
}
int nsg_debug_stamps_read(nsg_evaluator* ev, unsigned long long* dst) {
    if (!ev || !ev->stamps.p) return fail(NSG_E_INVALID, "stamps not enabled");
    NSG_HIP(hipDeviceSynchronize());
    NSG_HIP(hipMemcpy(dst, ev->stamps.p, ev->stamps.bytes, hipMemcpyDeviceToHost));
    return NSG_OK;
}
#endif

// (the CPU stand-in executors nsg_cpu_executor_* live in cpu_executor.cc: host-only code, built with the
// reference's release flags)

} // extern "C"
