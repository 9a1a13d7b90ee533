/*
 * nsg.h -- C ABI of the MI355X-native batched NN evaluator for nshogi-engine.
 *
 * This is the drop-in boundary (DESIGN.md "Boundary").  Every entry point is
 * `extern "C"`, takes plain pointers and sizes, and replaces one member of
 * the reference's executor plugin interface; the reference file:line each one
 * stands in for is cited next to it (paths under /root/reference/).
 * The C++ adapter `nshogi::engine::infer::Hip` (include/nshogi_engine_amd/
 * infer/hip.h) wraps this ABI behind the reference's `infer::Infer` virtuals
 * so src/mcts and src/selfplay link unchanged.
 *
 * Threading contract (same as the reference): one evaluator
 * per evaluation thread, at most one batch in flight per evaluator, not
 * thread-safe per object, no global mutable state in the HIP evaluator (one
 * exception below it, kept on purpose: the CPU stand-in Random executor shares
 * ONE function-static distribution object across all its instances, exactly as
 * random.cc:32 does -- a uniform_real_distribution<float> keeps no state between
 * draws with libstdc++, so outputs depend on each object's own engine only).
 * Results in the Dst* buffers are defined only after nsg_await() returns.
 *
 * All functions return NSG_OK (0) on success or a negative NSG_E_* code;
 * nsg_last_error() returns a thread-local description of the last failure.
 * There is NO CPU fallback: without a HIP device every compute entry fails.
 */
#ifndef NSG_H
#define NSG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSG_OK 0
#define NSG_E_INVALID (-1)   /* bad argument / wrong state                  */
#define NSG_E_HIP (-2)       /* a HIP runtime call failed                   */
#define NSG_E_IO (-3)        /* weight file could not be opened / read      */
#define NSG_E_FORMAT (-4)    /* weight blob malformed or shape mismatch     */
#define NSG_E_BUSY (-5)      /* a batch is already in flight                */
#define NSG_E_NOT_LOADED (-6) /* compute called before nsg_load*            */

#define NSG_NUM_SQUARES 81       /* core::NumSquares (trt.cc:60)            */
#define NSG_MOVE_INDEX_MAX 2187  /* ml::MoveIndexMax = 27*81 (trt.cc:62,205) */
#define NSG_BITBOARD_BYTES 16    /* sizeof(ml::FeatureBitboard)             */

/* Arithmetic the trunk computes in.  fp32 = exact f32 MFMA
 * (v_mfma_f32_16x16x4_f32); fp16/bf16 = 16-bit operands, f32 accumulate
 * (v_mfma_f32_16x16x32_{f16,bf16}).  The reference enables TF32
 * (trt.cc:161); gfx950 has no TF32 MFMA. */
#define NSG_PRECISION_FP32 0
#define NSG_PRECISION_FP16 1
#define NSG_PRECISION_BF16 2
/* f32-equivalent on the f16 matrix cores: values carried as (f16 hi, f16 lo)
 * pairs, products evaluated as hi*hi + lo*hi + hi*lo with f32 accumulation
 * (~22 significant bits). */
#define NSG_PRECISION_F16X3 3
/* F16X3 with the two correction products (each 2^-11 of the result) evaluated on
 * fp8 copies of the operands by the MX matrix instruction
 * (v_mfma_scale_f32_16x16x128_f8f6f4): 2.1 instead of 3 MFMA units per MAC,
 * ~2e-4 from the CPU fp32 executor on the 20x256 net (F16X3: 5e-6).  Trunk
 * convolutions only; the heads and the value MLP run as F16X3. */
#define NSG_PRECISION_F16M8 4
/* F16M8 with the correction operands in e2m3 (fp6) under one E8M0 scale per 32
 * input channels, applied by the MX instruction itself: with both operands in
 * fp6 it retires K = 128 in the cycles of one f16 MFMA, 1.5 MFMA units per MAC,
 * and block scales instead of F16M8's fixed ones mean no clamp window -- the
 * exponent follows the data.  Same accuracy class as F16M8 (~1e-4 on the
 * 20x256 net); trunk convolutions only, heads and value MLP run as F16X3. */
#define NSG_PRECISION_F16M6 5

typedef struct nsg_evaluator nsg_evaluator;

/* Replaces infer::TensorRT::TensorRT(int GPUId, uint16_t BatchSizeMax,
 * uint16_t NumChannels) -- src/infer/trt.h:44, src/infer/trt.cc:52-80:
 * binds the device, allocates the device-side input/plane/output buffers
 * sized for batch_size_max and creates one non-blocking stream. */
int nsg_create(int gpu_id, int batch_size_max, int num_channels,
               nsg_evaluator** out);

/* Replaces infer::TensorRT::~TensorRT() -- src/infer/trt.cc:82-107. */
int nsg_destroy(nsg_evaluator* ev);

/* Selects the trunk arithmetic (NSG_PRECISION_*).  Must be called before
 * nsg_load*.  An evaluator starts with NSG_PRECISION_FP32 unless the
 * environment variable NSG_PRECISION (0..5, or fp32 fp16 bf16 f16x3 f16m8
 * f16m6; anything else makes nsg_create fail) names another default -- so an
 * engine whose executor ladder only constructs and loads
 * can be run at the benchmarked arithmetic, NSG_PRECISION=f16m6, unchanged.
 * (The reference fixes this at engine-build time through BuilderFlag::kTF32,
 * src/infer/trt.cc:160-161.) */
int nsg_set_precision(nsg_evaluator* ev, int precision);

/* Replaces infer::TensorRT::load(const std::string& Path, bool) --
 * src/infer/trt.h:47, src/infer/trt.cc:109-232.  `path` names either the
 * ONNX model file the engine passes (trt.cc:121-131, default
 * ./res/model.onnx, src/context.h:93) or an NSGW v1 weight file (DESIGN.md
 * "Weight file"); the two are told apart by their first bytes.  An ONNX model
 * must be of the topology family this library runs (csrc/onnx_reader.h) and
 * obey the reference's tensor contract (input "input", outputs "policy",
 * "value", "draw", trt.cc:144-150,193-227); anything else fails with
 * NSG_E_FORMAT and a message naming the node (the reference's parser failure,
 * trt.cc:127-131).  BN is folded and the weights are re-laid into MFMA
 * fragment order on upload.  The policy width is checked against
 * NSG_MOVE_INDEX_MAX as trt.cc:193-210 does. */
int nsg_load(nsg_evaluator* ev, const char* path);
/* The ONNX -> NSGW v1 conversion nsg_load applies, on host memory and with no
 * device: writes *nsgw_size and, if dst != NULL, the blob (capacity bytes
 * available).  Call with dst = NULL first to size the buffer. */
int nsg_convert_onnx(const void* onnx, size_t size, void* dst, size_t capacity,
                     size_t* nsgw_size);
/* Same, from a host blob / from a device-resident blob (e.g. the buffer an
 * RCCL broadcast just filled). */
int nsg_load_memory(nsg_evaluator* ev, const void* blob, size_t size);
int nsg_load_device_blob(nsg_evaluator* ev, const void* device_blob,
                         size_t size);
/* Replaces the model re-read of every further executor: the reference builds
 * NumGPUs x threads-per-GPU executors that each load the file again
 * (src/selfplay/main.cc:189-195, src/mcts/manager.cc:168-179,
 * src/mcts/evaluationworker.cc:83-86).  Here `ev` adopts the network `src`
 * has loaded: on the same device both share one copy of the packed weights,
 * on another device `ev` takes a peer copy (hipMemcpyPeer over xGMI).  Same
 * precision and plane count required; `src` may be destroyed afterwards. */
int nsg_load_shared(nsg_evaluator* ev, nsg_evaluator* src);

/* Replaces infer::Infer::computeNonBlocking(const ml::FeatureBitboard*
 * Features, std::size_t BatchSize, float* DstPolicy, float* DstWinRate,
 * float* DstDrawRate) -- src/infer/infer.h:25-27, src/infer/trt.cc:234-272.
 * `features` = batch * num_channels 16-byte feature bitboards (host memory,
 * pinned or not); DstPolicy gets batch*2187 raw logits, DstWinRate/
 * DstDrawRate batch floats in [0,1].  Returns after enqueueing H2D ->
 * planes -> network -> 3x D2H on the evaluator's stream. */
int nsg_compute_nonblocking(nsg_evaluator* ev, const void* features,
                            size_t batch_size, float* dst_policy,
                            float* dst_win_rate, float* dst_draw_rate);

/* Replaces infer::Infer::computeBlocking -- src/infer/infer.h:28-30,
 * src/infer/trt.cc:274-279. */
int nsg_compute_blocking(nsg_evaluator* ev, const void* features,
                         size_t batch_size, float* dst_policy,
                         float* dst_win_rate, float* dst_draw_rate);

/* Extension (no reference counterpart): the evaluation with the
 * legal-move lookup moved to the device.  The caller sends, per position, the
 * policy indices of its legal moves -- what FeedWorker::feedResult
 * (src/mcts/feedworker.cc:120-127) and Frame::setEvaluation
 * (src/selfplay/frame.cc:105-118) look up one by one in the 2187-wide row -- and
 * receives only those values: D2H shrinks from 8748 B to ~4 B x legal moves per
 * position.  move_offsets has batch_size+1 entries (move_offsets[0] = 0, prefix
 * sums; at most 593 moves per position), move_indices move_offsets[batch_size]
 * entries in [0, 2187).  With softmax != 0 each position's values are
 * softmax(logits) (temperature 1, f32), i.e. the priors the host would compute.
 * dst_values gets move_offsets[batch_size] floats; same one-batch-in-flight and
 * await() contract as nsg_compute_nonblocking. */
int nsg_compute_gather_nonblocking(nsg_evaluator* ev, const void* features, size_t batch_size,
                                   const uint16_t* move_indices, const uint32_t* move_offsets,
                                   int softmax, float* dst_values, float* dst_win_rate,
                                   float* dst_draw_rate);
int nsg_compute_gather_blocking(nsg_evaluator* ev, const void* features, size_t batch_size,
                                const uint16_t* move_indices, const uint32_t* move_offsets,
                                int softmax, float* dst_values, float* dst_win_rate,
                                float* dst_draw_rate);
#define NSG_MAX_LEGAL_MOVES 593

/* Replaces infer::Infer::await() -- src/infer/infer.h:31, trt.cc:281-283
 * (cudaStreamSynchronize). */
int nsg_await(nsg_evaluator* ev);

/* Replaces infer::Infer::isComputing() -- src/infer/infer.h:32,
 * trt.cc:285-287 (cudaStreamQuery == cudaErrorNotReady).  Returns 1/0. */
int nsg_is_computing(nsg_evaluator* ev);

/* Replaces infer::TensorRT::resetGPU() -- src/infer/trt.h:57,
 * trt.cc:289-291: re-binds the calling thread to the evaluator's device. */
int nsg_reset_gpu(nsg_evaluator* ev);

/* Replaces cuda::extractBits<ChannelsFirst>(float* Dest, const uint64_t*
 * Src, int BatchSize, int NumChannels, cudaStream_t Stream) --
 * src/cuda/extractbit.h:21-23, src/cuda/extractbit.cu:76-96.  dst/src are
 * DEVICE pointers; `hip_stream` is a hipStream_t (NULL = default stream). */
int nsg_extract_bits(float* dst, const uint64_t* src, int batch_size,
                     int num_channels, int channels_first, void* hip_stream);

/* Page-locking for the caller-owned host batch buffers; replaces the
 * cudaHostRegister / cudaHostUnregister calls of evaluate::Evaluator --
 * src/evaluate/evaluator.cc:94-105 and :110-115. */
int nsg_host_register(void* ptr, size_t bytes);
int nsg_host_unregister(void* ptr);

/* ---- measurement hooks (no reference counterpart; used by bench.py and
 * tests to time the device-resident path and to read intermediates) ---- */

/* H2D of `batch_size` positions into the evaluator's device input buffer,
 * synchronous. */
int nsg_upload_features(nsg_evaluator* ev, const void* features,
                        size_t batch_size);
/* Enqueue planes + network on the already-resident input; outputs stay in
 * the evaluator's device buffers.  No host transfer, no sync. */
int nsg_forward_resident(nsg_evaluator* ev, size_t batch_size);
/* D2H of the device output buffers, synchronous. */
int nsg_download_outputs(nsg_evaluator* ev, size_t batch_size,
                         float* dst_policy, float* dst_win_rate,
                         float* dst_draw_rate);
/* Debug read-back of the trunk output as fp32 [batch][F][81] (NCHW). */
int nsg_download_trunk(nsg_evaluator* ev, size_t batch_size, float* dst);
/* Debug read-back of the trunk INPUT exactly as the last forward's plane
 * expansion wrote it: raw bytes, [batch][81][padded channels] in the trunk
 * precision's element layout; *row_bytes receives the
 * bytes per (board, square).  capacity is checked.  NSG_E_INVALID after a
 * forward of at most sixteen boards of a 256-channel net (the team trunk
 * decodes the bitboards inside its first layer: no plane buffer exists). */
int nsg_download_planes_raw(nsg_evaluator* ev, size_t batch_size, void* dst,
                            size_t capacity, size_t* row_bytes);

/* HIP-event timing of the plane-expansion kernel the forward pass actually runs (the bit selection of
 * cuda::extractBits, src/cuda/extractbit.cu:15-39, written straight into the trunk's input layout in the
 * evaluator's arithmetic): `iterations` launches on the resident input, on the evaluator's stream.
 * *bytes_per_launch = algorithmic bytes: batch x (num_channels x 16 read + 81 x padded channels x element size
 * written).  (Batches that run the team trunk decode the bitboards inside its first layer instead.) */
int nsg_time_planes(nsg_evaluator* ev, size_t batch_size, int iterations, float* avg_ms, double* bytes_per_launch);

/* HIP-event timing of the dominant kernel (the F->F 3x3 residual
 * convolution) on the evaluator's own stream.  While enabled, every forward
 * brackets its run of trunk-conv launches with two events; nsg_profile_read
 * synchronises and accumulates.  *launches counts kernel launches: 2 x blocks
 * per forward, or ONE per forward when the whole trunk (stem + 2 x blocks
 * convolutions) ran as one persistent launch (*trunk_launches == *forwards). */
int nsg_profile_enable(nsg_evaluator* ev, int enable);
int nsg_profile_read(nsg_evaluator* ev, double* trunk_ms_total,
                     uint64_t* trunk_launches, double* forward_ms_total,
                     uint64_t* forwards);

typedef struct nsg_info {
    int gpu_id;
    int batch_size_max;
    int num_channels;      /* input planes (86)                         */
    int channels;          /* trunk width F                             */
    int blocks;            /* residual blocks                           */
    int value_channels;
    int value_hidden;
    int precision;         /* NSG_PRECISION_*                           */
    int loaded;
    int compute_units;     /* hipDeviceProp_t.multiProcessorCount       */
    int clock_khz;         /* hipDeviceProp_t.clockRate                 */
    uint64_t param_count;
    double flops_per_position;       /* SURVEY.md 8d formula             */
    double trunk_conv_flops_per_position; /* one F->F 3x3 conv: 2*81*9*F*F */
    char device_name[128];
    /* Load-time estimate of the largest trunk activation (six standard deviations of the widest
     * channel, second moments pushed through the folded layers) and whether an F16M8 evaluator
     * found it outside the window its fixed-scale e4m3 copies cover (|x| < ~224): it then runs
     * every batch on its F16X3 copy of the trunk (f32-equivalent, slower).  F16M6 carries one
     * exponent per 32 channels and has no such window. */
    double activation_bound_estimate;
    int f16m8_window_fallback;
} nsg_info;
int nsg_get_info(nsg_evaluator* ev, nsg_info* info);

/* Host statistics of this evaluator since creation: forward passes enqueued and
 * positions in them -- average batch size = positions / batches, what the
 * reference accumulates in mcts::Statistics (src/mcts/statistics.h:74-98:
 * evaluationCount, batchSizeAccumulated) and prints after every search
 * (src/protocol/usilogger.cc:79-83,129-134).  The self-play driver prints its own
 * average batch size and cache-hit ratio (selfplayinfo.cc:51-57,72-78).
 * Profiler markers: with NSG_ROCTX=1 in the environment every forward pass wraps
 * its phases in roctx ranges nsg.h2d / nsg.planes / nsg.trunk / nsg.heads / nsg.d2h
 * for `rocprofv3 --marker-trace`. */
int nsg_get_stats(nsg_evaluator* ev, uint64_t* batches, uint64_t* positions);

/* The team trunk (batches of at most sixteen boards of a 256- or 192-channel net: every 3x3 layer in one
 * persistent launch whose workgroups hand activations to each other) needs all its workgroups resident at once.
 * It is taken only when boards x workgroups-per-board fits the device's compute units, by one process per device
 * (an advisory file lock named after the PCI bus id) and one launch per device at a time inside that process.
 * Should a launch still wait in vain (~1 s), the batch is re-run on the per-layer kernels before nsg_await or any
 * other synchronising entry returns -- the call succeeds -- and the evaluator keeps to those kernels afterwards.
 * *enabled: 1 the team path is in use, 0 it is off (no layer list, NSG_TEAM_TRUNK=0, or after a fallback),
 * -1 another process holds the device's token; *members_last: workgroups per board of the most recent team
 * launch; *fallbacks: launches that gave up and were re-run. */
int nsg_get_team_stats(nsg_evaluator* ev, int* enabled, int* members_last, uint64_t* fallbacks);
/* How the most recent forward pass launched its trunk: 0 = per-layer kernels (or the persistent launch whose workgroups
 * own whole boards: no hand-off), 1 = the team trunk, 2 = the cooperative trunk -- the two-way K split of 65 ... CUs/2
 * boards of a 256-channel F16M6 net as ONE launch whose two workgroups per board hand their channel halves to each
 * other (same residency rule, device token, give-up and re-run as the team trunk; NSG_COOP_TRUNK=0 switches it off).
 * *coop_enabled: 1 in use, 0 off (not applicable, switched off, or after a fallback), -1 another process holds the
 * device's token. */
int nsg_get_last_launch_kind(nsg_evaluator* ev, int* kind, int* coop_enabled);

/* Launch plan of the most recent forward pass (tests and tuning): boards per
 * workgroup, 16-channel fragments per wave, waves per workgroup, and the number
 * of independent launch chains the batch ran as -- half-batch chains of one
 * plan (more tiles than CUs), or a full part plus remainder parts with their own
 * plans (batch sizes just above the ones a plan fills the chip at; the reported
 * plan is the first part's).  All zero before the first forward pass. */
int nsg_get_last_plan(nsg_evaluator* ev, int* boards_per_group, int* fragments_per_wave,
                      int* waves_per_group, int* chains);
/* How the waves of one channel group shared a one-board tile in the most recent forward pass:
 * row_split waves took disjoint row fragments (small tiles), or k_split waves took disjoint ranges
 * of the input channels for all rows and summed their accumulators (F16M8 / F16M6 at small and mid batches);
 * with k_split = 4 a row_split of 2, 3 or 6 means that many WORKGROUPS per channel group, each on its
 * share of the rows (the smallest batches).  1 / 1 for ordinary tiles, 0 / 0 before the first pass. */
int nsg_get_last_split(nsg_evaluator* ev, int* row_split, int* k_split);
/* ... and how many waves of a channel group shared every chunk pair's SLABS between them (two-board MX tiles at
 * mid batches: 4 up to CUs/2 boards, 2 up to CUs boards); 1 for every other plan, 0 before the first pass. */
int nsg_get_last_slab_split(nsg_evaluator* ev, int* slab_split);
/* Arithmetic the most recent forward pass ran its trunk in (NSG_PRECISION_*; -1 before
 * the first pass).  An F16M8 evaluator runs small batches for which it has no F16M8 tile plan
 * (channel counts other than 256) as F16X3. */
int nsg_get_last_trunk_precision(nsg_evaluator* ev, int* precision);

/* CPU stand-in executors of the reference (src/infer/zero.cc, nothing.cc,
 * random.cc): product code, selectable like EXECUTOR=zero|nothing|random
 * (Makefile:107-121).  kind: 0 = Zero, 1 = Nothing, 2 = Random(seed). */
typedef struct nsg_cpu_executor nsg_cpu_executor;
int nsg_cpu_executor_create(int kind, uint64_t seed, nsg_cpu_executor** out);
int nsg_cpu_executor_destroy(nsg_cpu_executor* ex);
int nsg_cpu_executor_compute(nsg_cpu_executor* ex, const void* features,
                             size_t batch_size, float* dst_policy,
                             float* dst_win_rate, float* dst_draw_rate);

const char* nsg_last_error(void);
const char* nsg_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NSG_H */
