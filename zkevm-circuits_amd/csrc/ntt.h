// NTT plan / launcher interface (internal to libzkhip).
#pragma once
#include "common.h"

namespace zkhip {

struct NttPlan {
    uint32_t log_n = 0;
    uint64_t omega[4] = {0, 0, 0, 0};
    uint32_t h = 0;              // low table holds omega^i for i < 2^h
    void* d_lo = nullptr;
    void* d_hi = nullptr;
    void* d_lo_w = nullptr;      // the same tables in the 2^261 Montgomery domain (transforms of 2^11 points and more)
    void* d_hi_w = nullptr;
    // twiddles of pass boundary p by one lookup instead of lo * hi: d_bt[p][k] = omega^(k * 2^(log_n - bits_p)) for
    // k < 2^bits_p, bits_p = r_0 + ... + r_(p+1), 2^261 domain.  Built up to 2^22 entries (128 MiB; the last boundary
    // of a 2^21 transform is a 64 MiB table shared by every column of a batch): one product per element saved for
    // 32 more bytes read.  Null beyond that size.
    std::vector<void*> d_bt;
    std::vector<uint32_t> radix; // log2 radix of each pass
    ~NttPlan();
};

struct NttFuse {
    uint32_t pre_mode = 0;    // 1: input i multiplied by pre[i % 3]
    uint32_t post_mode = 0;   // 1: output multiplied by post[0]; 2: output i multiplied by post[i % 3]
    fe pre[3];
    fe post[3];
    // 1: the output goes out split by the parity of its index: element i of the N results is stored at
    // (i & 1) * N/2 + (i >> 1), i.e. the even-indexed results first, then the odd-indexed ones (needs n_out = N)
    uint32_t split_out = 0;
};

int ntt_plan_create(zkhip_ctx* ctx, uint32_t log_n, const uint64_t omega[4], NttPlan** out);
int ntt_run(zkhip_ctx* ctx, const NttPlan* plan, const fe* src, size_t src_stride, uint32_t n_in, fe* dst,
            size_t dst_stride, uint32_t n_out, size_t batch, const NttFuse& fuse, fe* tmp0, fe* tmp1);
int scale_periodic(zkhip_ctx* ctx, fe* a, const fe* d_scale, uint32_t period, size_t n);
int ntt_init(zkhip_ctx* ctx);

}  // namespace zkhip
