// Shared host-side declarations of libzkhip (context, error plumbing, timing hooks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include "../../include/zkhip.h"
#include "field.hip.h"

namespace zkhip {

struct NttPlan;
struct EvalhProgram;

struct TimingSlot {
    double total_us = 0;
    uint64_t launches = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace zkhip

struct zkhip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    std::string last_error;
    uint32_t next_bases_id = 1;
    // scratch arena reused across calls (grown on demand, never shrunk)
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    int timing = 0;   // 0 off, 1 every family, 2 only the dominant kernels ("msm_accumulate", "ntt")
    std::map<std::string, zkhip::TimingSlot> timers;
    int num_cus = 256;
    std::vector<zkhip::NttPlan*> plans;   // cached per (log_n, omega)
    // program of the last evaluate_h call: instructions, multiplications, column reads, value slots
    uint64_t evalh_stats[4] = {0, 0, 0, 0};
    // evaluate_h programs by shape (hash of the graphs and counts), built once like upstream's `ProvingKey::ev`; the
    // linked program, scalars, constants and spilled slots of a call live in grow-only device buffers
    std::map<std::pair<uint64_t, uint64_t>, std::shared_ptr<zkhip::EvalhProgram>> evalh_cache;
    uint64_t evalh_builds = 0, evalh_cache_hits = 0;
    void* evalh_buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t evalh_buf_bytes[4] = {0, 0, 0, 0};
    std::vector<uint32_t> evalh_linked;
    bool evalh_attr_set = false;
    // transfer stream of zkhip_upload_overlapped: copies run beside the compute stream until zkhip_upload_fence joins them
    hipStream_t upload_stream = nullptr;
    hipEvent_t ev_upload_base = nullptr, ev_upload_done = nullptr;
    bool upload_open = false;             // copies were issued since the last fence
};

namespace zkhip {

int set_error(zkhip_ctx* ctx, int code, const char* fmt, ...);
int ensure_scratch(zkhip_ctx* ctx, size_t bytes);

#define ZK_HIP(ctx, expr)                                                                             \
    do {                                                                                              \
        hipError_t _e = (expr);                                                                       \
        if (_e != hipSuccess)                                                                         \
            return zkhip::set_error((ctx), _e == hipErrorOutOfMemory ? ZKHIP_ERR_OOM : ZKHIP_ERR_HIP, \
                                    "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define ZK_CHECK(ctx, cond, ...)                                                  \
    do {                                                                          \
        if (!(cond)) return zkhip::set_error((ctx), ZKHIP_ERR_INVALID, __VA_ARGS__); \
    } while (0)

// RAII-less timing bracket: records HIP events around a kernel family when timing is enabled.
struct TimedRegion {
    zkhip_ctx* ctx;
    const char* family;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipStream_t stream = nullptr;
    // family == nullptr or timing disabled: no-op.  s == nullptr: the context's stream.
    TimedRegion(zkhip_ctx* c, const char* f, hipStream_t s = nullptr);
    ~TimedRegion();
};

// ---- host-side field helpers (4 x u64 Montgomery) used for constants and final folds --------------
struct HostField {
    uint64_t p[4], r[4], r2[4], inv;
};
extern const HostField HFR, HFQ;
void h_mul(const HostField& F, uint64_t o[4], const uint64_t a[4], const uint64_t b[4]);
void h_add(const HostField& F, uint64_t o[4], const uint64_t a[4], const uint64_t b[4]);
void h_sub(const HostField& F, uint64_t o[4], const uint64_t a[4], const uint64_t b[4]);
void h_inv(const HostField& F, uint64_t o[4], const uint64_t a[4]);
void h_pow(const HostField& F, uint64_t o[4], const uint64_t a[4], uint64_t e);
void h_from_u64(const HostField& F, uint64_t o[4], uint64_t v);
bool h_is_zero(const uint64_t a[4]);
bool h_eq(const uint64_t a[4], const uint64_t b[4]);
// Jacobian G1 on the host (final window fold / partial-sum fold only: O(10^2..10^3) operations per MSM)
void h_g1_double(uint64_t o[12], const uint64_t p[12]);
void h_g1_add(uint64_t o[12], const uint64_t p[12], const uint64_t q[12]);
void h_g1_to_affine(uint64_t o[8], const uint64_t p[12]);

}  // namespace zkhip
