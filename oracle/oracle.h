/*
 * oracle.h -- CPU restatement of the nshogi-engine NN-evaluation hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load or call anything under oracle/.  The product path (libnsg.so) never
 * links, loads or falls back to this library.
 *
 * Parity pinning status (see DESIGN.md "Oracle"):
 *   - extract bits (K1/K2): restated line by line from
 *     /root/reference/src/cuda/extractbit.cu:15-39 (NCHW) and :41-68 (NHWC).
 *     The reference holds no golden vectors for it (its only test,
 *     src/test/test_extractbit.cc:26-91, derives expectations from the
 *     un-vendored libnshogi at run time), and the reference cannot be built
 *     here (needs libnshogi + CUDA).  Pinned by synthetic branch-coverage
 *     vectors committed under tests/golden/ (SURVEY.md 8c G2).
 *   - infer::Random / Zero / Nothing: restated from
 *     /root/reference/src/infer/random.cc:28-42, zero.cc:25-31,
 *     nothing.cc:22-24.  Pinned by the known-answer vector G1 (first raw
 *     std::mt19937_64(0) output 2947667278772165694; first four policy floats
 *     0x3e23a0df 0x3f7dfd3a 0x3d221321 0x3f18f569) and cross-checked against
 *     this image's libstdc++ <random> in tests/test_oracle_executors.py.
 *   - policy/value/draw network: PARITY UNPINNED.  The reference ships no
 *     model, no topology and no expected outputs (it loads an arbitrary ONNX
 *     through TensorRT, src/infer/trt.cc:109-232).  The network oracle below
 *     defines the build's own topology (DESIGN.md "Network") in plain fp32
 *     storage with fp64 accumulation and is the checker for the HIP kernels.
 */
#ifndef NSG_ORACLE_H
#define NSG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NSG_ORACLE_NUM_SQUARES 81
#define NSG_ORACLE_MOVE_INDEX_MAX 2187 /* 27 * 81, ml::MoveIndexMax (trt.cc:205) */

/* ---- a6: feature bitboards -> planes (extractbit.cu) ---- */
/* NCHW: dst[(b*C + c)*81 + bit]   (extractbit.cu:15-39)  */
void nsg_oracle_extract_bits_nchw(float* dst, const uint64_t* src, int batch,
                                  int channels);
/* NHWC: dst[(b*81 + bit)*C + c]   (extractbit.cu:41-68)  */
void nsg_oracle_extract_bits_nhwc(float* dst, const uint64_t* src, int batch,
                                  int channels);

/* ---- a8: CPU stand-in executors (random.cc / zero.cc / nothing.cc) ---- */
typedef struct nsg_oracle_mt19937_64 {
    uint64_t mt[312];
    int idx;
} nsg_oracle_mt19937_64;

void nsg_oracle_mt_seed(nsg_oracle_mt19937_64* st, uint64_t seed);
uint64_t nsg_oracle_mt_next(nsg_oracle_mt19937_64* st);
/* std::uniform_real_distribution<float>(0,1)(rng) as libstdc++ computes it */
float nsg_oracle_uniform01f(nsg_oracle_mt19937_64* st);

/* infer::Random::computeNonBlocking (random.cc:28-42) */
void nsg_oracle_random_compute(nsg_oracle_mt19937_64* st, size_t batch,
                               float* policy, float* win, float* draw);
/* infer::Zero::computeNonBlocking (zero.cc:25-31) */
void nsg_oracle_zero_compute(size_t batch, float* policy, float* win,
                             float* draw);

/* ---- a7: the policy/value/draw network (build-defined topology) ---- */
typedef struct nsg_oracle_net {
    int in_channels;    /* 86 */
    int channels;       /* F */
    int blocks;         /* N residual blocks */
    int policy_channels; /* 27 */
    int value_channels; /* VC */
    int value_hidden;   /* VH */
    float bn_eps;
    /* All tensors in PyTorch layout, fp32, NOT folded. */
    const float* stem_w;            /* [F][Cin][3][3] */
    const float* stem_bn;           /* [4][F]: gamma, beta, mean, var */
    const float* const* block_w1;   /* blocks x [F][F][3][3] */
    const float* const* block_bn1;  /* blocks x [4][F] */
    const float* const* block_w2;
    const float* const* block_bn2;
    const float* policy_w;          /* [27][F] */
    const float* policy_b;          /* [27] */
    const float* value_w;           /* [VC][F] */
    const float* value_bn;          /* [4][VC] */
    const float* fc1_w;             /* [VH][VC*81] */
    const float* fc1_b;             /* [VH] */
    const float* fc2_w;             /* [2][VH]  row0 = value, row1 = draw */
    const float* fc2_b;             /* [2] */
} nsg_oracle_net;

/* planes: [batch][Cin][81] fp32 (NCHW).  Outputs as the reference executor
 * contract (trt.cc:193-227): policy [batch][2187] raw logits with index
 * c*81+sq, value in [0,1], draw in [0,1].
 * If trunk_out != NULL it receives the trunk output [batch][F][81]. */
void nsg_oracle_net_forward(const nsg_oracle_net* net, const float* planes,
                            int batch, float* policy, float* value,
                            float* draw, float* trunk_out);

/* Convenience: parse an NSGW v1 blob (DESIGN.md "Weight file") that stays
 * alive in memory; fills `net` with pointers into the blob.  Returns 0 on
 * success.  `ptr_storage` must hold 4*blocks pointers. */
int nsg_oracle_net_from_blob(const void* blob, size_t size, nsg_oracle_net* net,
                             const float** ptr_storage, size_t ptr_capacity);

/* Whole path: bitboards -> planes (NCHW) -> network. */
void nsg_oracle_evaluate(const nsg_oracle_net* net, const uint64_t* bitboards,
                         int batch, float* policy, float* value, float* draw);

#ifdef __cplusplus
}
#endif
#endif /* NSG_ORACLE_H */
