// cpu_executor.cc -- the reference's CPU stand-in executors behind the C ABI (nsg.h: nsg_cpu_executor_*):
// infer::Zero (src/infer/zero.cc:25-31), infer::Nothing (nothing.cc:22-24), infer::Random (random.cc:21-42).
// Host-only code: compiled by the host compiler with the flags the reference's release build gives these files
// (Makefile:30-32,155-158: -O3 -ffast-math -fno-rtti -fno-stack-protector ... and the AVX2 set), so that the CPU
// baseline bench.py times is the reference's EXECUTOR=random path as the reference would build it.  (-flto is left
// out: this is one translation unit linked into a library the device compiler links.)
#include "../../include/nsg.h"

#include <cstring>
#include <random>

struct nsg_cpu_executor {
    int kind;
    std::mt19937_64 rng; // random.h:41
};

extern "C" {

int nsg_cpu_executor_create(int kind, uint64_t seed, nsg_cpu_executor** out) {
    if (!out || kind < 0 || kind > 2) return NSG_E_INVALID;
    *out = new nsg_cpu_executor{kind, std::mt19937_64(seed)}; // random.cc:21-23
    return NSG_OK;
}

int nsg_cpu_executor_destroy(nsg_cpu_executor* ex) {
    delete ex;
    return NSG_OK;
}

int nsg_cpu_executor_compute(nsg_cpu_executor* ex, const void*, size_t batch_size,
                             float* dst_policy, float* dst_win_rate, float* dst_draw_rate) {
    if (!ex) return NSG_E_INVALID;
    if (ex->kind == 0) { // zero.cc:25-31
        memset(dst_policy, 0, batch_size * NSG_MOVE_INDEX_MAX * sizeof(float));
        memset(dst_win_rate, 0, batch_size * sizeof(float));
        memset(dst_draw_rate, 0, batch_size * sizeof(float));
    } else if (ex->kind == 2) { // random.cc:28-42
        // one distribution object shared by every executor, as the
        // function-static of random.cc:32 is
        static std::uniform_real_distribution<float> distribution(0, 1);
        for (size_t i = 0; i < batch_size; ++i) {
            for (size_t j = 0; j < NSG_MOVE_INDEX_MAX; ++j)
                dst_policy[i * NSG_MOVE_INDEX_MAX + j] = distribution(ex->rng);
            dst_win_rate[i] = distribution(ex->rng);
            dst_draw_rate[i] = distribution(ex->rng);
        }
    } // kind 1 = Nothing: nothing.cc:22-24
    return NSG_OK;
}

} // extern "C"
