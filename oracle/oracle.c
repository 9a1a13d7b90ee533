/*
 * oracle.c -- CPU restatement of the nshogi-engine NN-evaluation hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h for the rules and the parity
 * pinning status of each function).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------
 * a6: extract bits.  Follows /root/reference/src/cuda/extractbit.cu.
 * One "thread" of the reference kernel = one (Index, BitIndex) pair.
 * ---------------------------------------------------------------------- */
static inline int32_t extract_one(uint64_t lo, uint64_t hi, int bit_index) {
    /* extractbit.cu:20-21 */
    const int rotate = (int)((hi >> 24) & 1);
    const int value = (int)(hi >> 32);
    /* extractbit.cu:26 */
    const int target_square = bit_index * (1 - 2 * rotate) + 80 * rotate;
    /* extractbit.cu:30 */
    const int shift_amount = target_square - 63 * (target_square >= 63);
    /* extractbit.cu:34 */
    const uint64_t word = (target_square >= 63) ? hi : lo;
    /* extractbit.cu:36-37: uint64 * int -> uint64, truncated to int */
    const uint64_t mask = 1ULL << shift_amount;
    const uint64_t prod = ((word & mask) >> shift_amount) * (uint64_t)(int64_t)value;
    return (int32_t)(uint32_t)prod;
}

void nsg_oracle_extract_bits_nchw(float* dst, const uint64_t* src, int batch,
                                  int channels) {
    int32_t* d = (int32_t*)dst; /* extractbit.cu:79-84: int alias of floats */
    for (int index = 0; index < batch * channels; ++index) {
        const uint64_t lo = src[2 * index];
        const uint64_t hi = src[2 * index + 1];
        for (int bit = 0; bit < 81; ++bit) {
            d[(size_t)index * 81 + bit] = extract_one(lo, hi, bit);
        }
    }
}

void nsg_oracle_extract_bits_nhwc(float* dst, const uint64_t* src, int batch,
                                  int channels) {
    int32_t* d = (int32_t*)dst;
    for (int b = 0; b < batch; ++b) {
        for (int bit = 0; bit < 81; ++bit) {
            for (int c = 0; c < channels; ++c) {
                const uint64_t lo = src[2 * ((size_t)b * channels + c)];
                const uint64_t hi = src[2 * ((size_t)b * channels + c) + 1];
                /* extractbit.cu:65-66 */
                d[((size_t)b * 81 + bit) * channels + c] =
                    extract_one(lo, hi, bit);
            }
        }
    }
}

/* ------------------------------------------------------------------------
 * a8: std::mt19937_64 + std::uniform_real_distribution<float>(0,1).
 * mt19937_64 is the ISO C++ [rand.predef] engine (n=312, m=156, r=31,
 * a=0xB5026F5AA96619E9, u=29, d=0x5555555555555555, s=17,
 * b=0x71D67FFFEDA60000, t=37, c=0xFFF7EEE000000000, l=43,
 * f=6364136223846793005).
 * ---------------------------------------------------------------------- */
void nsg_oracle_mt_seed(nsg_oracle_mt19937_64* st, uint64_t seed) {
    st->mt[0] = seed;
    for (int i = 1; i < 312; ++i) {
        st->mt[i] = 6364136223846793005ULL *
                        (st->mt[i - 1] ^ (st->mt[i - 1] >> 62)) +
                    (uint64_t)i;
    }
    st->idx = 312;
}

uint64_t nsg_oracle_mt_next(nsg_oracle_mt19937_64* st) {
    if (st->idx >= 312) {
        const uint64_t upper = 0xFFFFFFFF80000000ULL;
        const uint64_t lower = 0x000000007FFFFFFFULL;
        for (int i = 0; i < 312; ++i) {
            const uint64_t x =
                (st->mt[i] & upper) | (st->mt[(i + 1) % 312] & lower);
            uint64_t xa = x >> 1;
            if (x & 1ULL) {
                xa ^= 0xB5026F5AA96619E9ULL;
            }
            st->mt[i] = st->mt[(i + 156) % 312] ^ xa;
        }
        st->idx = 0;
    }
    uint64_t y = st->mt[st->idx++];
    y ^= (y >> 29) & 0x5555555555555555ULL;
    y ^= (y << 17) & 0x71D67FFFEDA60000ULL;
    y ^= (y << 37) & 0xFFF7EEE000000000ULL;
    y ^= (y >> 43);
    return y;
}

/* libstdc++ std::generate_canonical<float, 24>(urng) with a 64-bit engine:
 * k = max(1, ceil(24/64)) = 1 draw; sum = float(urng() - min) * 1.0f;
 * ret = sum / float(2^64); if (ret >= 1) ret = nextafter(1.0f, 0.0f).
 * uniform_real_distribution<float>(0,1) returns ret * (1 - 0) + 0.
 * (Golden vector G1 of SURVEY.md 8c pins this.) */
float nsg_oracle_uniform01f(nsg_oracle_mt19937_64* st) {
    const float sum = (float)nsg_oracle_mt_next(st);
    const float range = 18446744073709551616.0f; /* 2^64 */
    float ret = sum / range;
    if (ret >= 1.0f) {
        ret = nextafterf(1.0f, 0.0f);
    }
    return ret;
}

void nsg_oracle_random_compute(nsg_oracle_mt19937_64* st, size_t batch,
                               float* policy, float* win, float* draw) {
    /* random.cc:34-41 */
    for (size_t i = 0; i < batch; ++i) {
        for (size_t j = 0; j < NSG_ORACLE_MOVE_INDEX_MAX; ++j) {
            policy[i * NSG_ORACLE_MOVE_INDEX_MAX + j] =
                nsg_oracle_uniform01f(st);
        }
        win[i] = nsg_oracle_uniform01f(st);
        draw[i] = nsg_oracle_uniform01f(st);
    }
}

void nsg_oracle_zero_compute(size_t batch, float* policy, float* win,
                             float* draw) {
    /* zero.cc:28-30 */
    memset(policy, 0, batch * NSG_ORACLE_MOVE_INDEX_MAX * sizeof(float));
    memset(win, 0, batch * sizeof(float));
    memset(draw, 0, batch * sizeof(float));
}

/* ------------------------------------------------------------------------
 * a7: the network.  Topology is the build's own (the reference only fixes
 * the tensor contract, trt.cc:144-150,193-227):
 *   stem   : conv3x3(Cin->F, no bias) -> BN -> ReLU
 *   block  : y = ReLU(BN1(conv3x3(x))); y = BN2(conv3x3(y)); x = ReLU(x + y)
 *   policy : conv1x1(F->27) + bias, logits index = c*81 + sq
 *   value  : v = ReLU(BN(conv1x1(F->VC))) flattened c*81+sq;
 *            h = ReLU(fc1 v + b1); o = fc2 h + b2;
 *            value = (tanh(o[0]) + 1)/2, draw = 1/(1+exp(-o[1])).
 * Storage fp32, accumulation fp64, BN applied un-folded.
 * ---------------------------------------------------------------------- */
static void conv3x3_bn(const float* in, int cin, const float* w,
                       const float* bn, float eps, int cout,
                       const float* residual, int relu, float* out) {
    /* in [cin][81], w [cout][cin][3][3], out [cout][81] */
    for (int oc = 0; oc < cout; ++oc) {
        double acc[81];
        for (int s = 0; s < 81; ++s) acc[s] = 0.0;
        for (int ic = 0; ic < cin; ++ic) {
            const float* wk = w + ((size_t)oc * cin + ic) * 9;
            const float* ip = in + (size_t)ic * 81;
            for (int ky = 0; ky < 3; ++ky) {
                for (int kx = 0; kx < 3; ++kx) {
                    const double wv = (double)wk[ky * 3 + kx];
                    const int dy = ky - 1, dx = kx - 1;
                    const int y0 = dy < 0 ? 1 : 0, y1 = dy > 0 ? 8 : 9;
                    const int x0 = dx < 0 ? 1 : 0, x1 = dx > 0 ? 8 : 9;
                    for (int y = y0; y < y1; ++y) {
                        for (int x = x0; x < x1; ++x) {
                            acc[y * 9 + x] +=
                                wv * (double)ip[(y + dy) * 9 + (x + dx)];
                        }
                    }
                }
            }
        }
        const double gamma = bn[0 * cout + oc], beta = bn[1 * cout + oc];
        const double mean = bn[2 * cout + oc], var = bn[3 * cout + oc];
        const double inv = gamma / sqrt(var + (double)eps);
        for (int s = 0; s < 81; ++s) {
            double v = (acc[s] - mean) * inv + beta;
            if (residual) v += (double)residual[(size_t)oc * 81 + s];
            if (relu && v < 0.0) v = 0.0;
            out[(size_t)oc * 81 + s] = (float)v;
        }
    }
}

void nsg_oracle_net_forward(const nsg_oracle_net* net, const float* planes,
                            int batch, float* policy, float* value,
                            float* draw, float* trunk_out) {
    const int F = net->channels, Cin = net->in_channels;
    const int VC = net->value_channels, VH = net->value_hidden;
    const int PC = net->policy_channels;
    float* x = (float*)malloc(sizeof(float) * (size_t)F * 81);
    float* y = (float*)malloc(sizeof(float) * (size_t)F * 81);
    float* z = (float*)malloc(sizeof(float) * (size_t)F * 81);
    float* v = (float*)malloc(sizeof(float) * (size_t)VC * 81);
    float* h = (float*)malloc(sizeof(float) * (size_t)VH);

    for (int b = 0; b < batch; ++b) {
        const float* in = planes + (size_t)b * Cin * 81;
        conv3x3_bn(in, Cin, net->stem_w, net->stem_bn, net->bn_eps, F, NULL, 1,
                   x);
        for (int k = 0; k < net->blocks; ++k) {
            conv3x3_bn(x, F, net->block_w1[k], net->block_bn1[k], net->bn_eps,
                       F, NULL, 1, y);
            conv3x3_bn(y, F, net->block_w2[k], net->block_bn2[k], net->bn_eps,
                       F, x, 1, z);
            float* t = x;
            x = z;
            z = t;
        }
        if (trunk_out) {
            memcpy(trunk_out + (size_t)b * F * 81, x,
                   sizeof(float) * (size_t)F * 81);
        }
        /* policy head */
        for (int c = 0; c < PC; ++c) {
            for (int s = 0; s < 81; ++s) {
                double acc = (double)net->policy_b[c];
                for (int ic = 0; ic < F; ++ic) {
                    acc += (double)net->policy_w[(size_t)c * F + ic] *
                           (double)x[(size_t)ic * 81 + s];
                }
                policy[(size_t)b * (PC * 81) + (size_t)c * 81 + s] =
                    (float)acc;
            }
        }
        /* value / draw head */
        for (int c = 0; c < VC; ++c) {
            const double gamma = net->value_bn[0 * VC + c];
            const double beta = net->value_bn[1 * VC + c];
            const double mean = net->value_bn[2 * VC + c];
            const double var = net->value_bn[3 * VC + c];
            const double inv = gamma / sqrt(var + (double)net->bn_eps);
            for (int s = 0; s < 81; ++s) {
                double acc = 0.0;
                for (int ic = 0; ic < F; ++ic) {
                    acc += (double)net->value_w[(size_t)c * F + ic] *
                           (double)x[(size_t)ic * 81 + s];
                }
                double r = (acc - mean) * inv + beta;
                v[(size_t)c * 81 + s] = (float)(r < 0.0 ? 0.0 : r);
            }
        }
        for (int j = 0; j < VH; ++j) {
            double acc = (double)net->fc1_b[j];
            const float* wr = net->fc1_w + (size_t)j * VC * 81;
            for (int i = 0; i < VC * 81; ++i) {
                acc += (double)wr[i] * (double)v[i];
            }
            h[j] = (float)(acc < 0.0 ? 0.0 : acc);
        }
        double o[2];
        for (int r = 0; r < 2; ++r) {
            double acc = (double)net->fc2_b[r];
            for (int j = 0; j < VH; ++j) {
                acc += (double)net->fc2_w[(size_t)r * VH + j] * (double)h[j];
            }
            o[r] = acc;
        }
        value[b] = (float)(0.5 * (tanh(o[0]) + 1.0));
        draw[b] = (float)(1.0 / (1.0 + exp(-o[1])));
    }
    free(x);
    free(y);
    free(z);
    free(v);
    free(h);
}

/* NSGW v1 blob: 64-byte header then fp32 tensors (DESIGN.md "Weight file"). */
int nsg_oracle_net_from_blob(const void* blob, size_t size, nsg_oracle_net* net,
                             const float** ptr_storage, size_t ptr_capacity) {
    if (size < 64) return -1;
    const unsigned char* p = (const unsigned char*)blob;
    if (memcmp(p, "NSGW", 4) != 0) return -2;
    uint32_t hdr[15];
    memcpy(hdr, p + 4, sizeof(hdr));
    if (hdr[0] != 1) return -3;
    net->in_channels = (int)hdr[1];
    net->channels = (int)hdr[2];
    net->blocks = (int)hdr[3];
    net->policy_channels = (int)hdr[4];
    net->value_channels = (int)hdr[5];
    net->value_hidden = (int)hdr[6];
    memcpy(&net->bn_eps, &hdr[7], 4);
    if (ptr_capacity < (size_t)4 * net->blocks) return -4;
    const size_t F = net->channels, Cin = net->in_channels;
    const size_t VC = net->value_channels, VH = net->value_hidden;
    const size_t PC = net->policy_channels;
    size_t need = F * Cin * 9 + 4 * F +
                  (size_t)net->blocks * 2 * (F * F * 9 + 4 * F) + PC * F + PC +
                  VC * F + 4 * VC + VH * VC * 81 + VH + 2 * VH + 2;
    if (size != 64 + need * sizeof(float)) return -5;
    const float* f = (const float*)(p + 64);
    net->stem_w = f; f += F * Cin * 9;
    net->stem_bn = f; f += 4 * F;
    const float** w1 = ptr_storage;
    const float** b1 = ptr_storage + net->blocks;
    const float** w2 = ptr_storage + 2 * net->blocks;
    const float** b2 = ptr_storage + 3 * net->blocks;
    for (int k = 0; k < net->blocks; ++k) {
        w1[k] = f; f += F * F * 9;
        b1[k] = f; f += 4 * F;
        w2[k] = f; f += F * F * 9;
        b2[k] = f; f += 4 * F;
    }
    net->block_w1 = w1;
    net->block_bn1 = b1;
    net->block_w2 = w2;
    net->block_bn2 = b2;
    net->policy_w = f; f += PC * F;
    net->policy_b = f; f += PC;
    net->value_w = f; f += VC * F;
    net->value_bn = f; f += 4 * VC;
    net->fc1_w = f; f += VH * VC * 81;
    net->fc1_b = f; f += VH;
    net->fc2_w = f; f += 2 * VH;
    net->fc2_b = f; f += 2;
    return 0;
}

void nsg_oracle_evaluate(const nsg_oracle_net* net, const uint64_t* bitboards,
                         int batch, float* policy, float* value, float* draw) {
    float* planes = (float*)malloc(sizeof(float) * (size_t)batch *
                                   net->in_channels * 81);
    nsg_oracle_extract_bits_nchw(planes, bitboards, batch, net->in_channels);
    nsg_oracle_net_forward(net, planes, batch, policy, value, draw, NULL);
    free(planes);
}
