// Shared helpers for the gfx950 kernels of libcst_hip.so (internal header, not the C ABI:
// the ABI is include/cst_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define CST_OK 0
#define CST_ERR_ARG 1
#define CST_ERR_LAUNCH 2

void cst_set_error(const char* fmt, ...);

#define CST_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            cst_set_error(__VA_ARGS__);        \
            return CST_ERR_ARG;                \
        }                                      \
    } while (0)

#define CST_LAUNCH_CHECK(name)                                                    \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            cst_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return CST_ERR_LAUNCH;                                                \
        }                                                                         \
    } while (0)

#define CST_WAVE 64

// Function attributes (the dynamic-LDS limit) are per DEVICE: a launcher sets them the first time it runs on each device of the
// process, not once per process.  `static CstPerDevice done; if (cst_first_on_device(done)) hipFuncSetAttribute(...)`.
struct CstPerDevice { bool done[64] = {}; };
static inline bool cst_first_on_device(CstPerDevice& f) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    if (f.done[dev]) return false;
    f.done[dev] = true;
    return true;
}

// Zero n 4-byte words with a KERNEL (csrc/pointwise.hip).  Never hipMemsetAsync inside a library call: under segmented
// hipGraph capture (capture_error_mode thread_local, needed next to RCCL's threads) a memset issued by the autograd engine
// thread was replayed out of order -- the column-sum target of the last encoder layer got zeroed BEFORE an earlier tenant of
// the same pool block wrote it, and 5.8e25 came back as a bias gradient (round-2 debugging, tools/debug/bucketed_dbg.py).
// Kernel nodes keep their stream order.
int cst_zero_words(void* p, long n_words, hipStream_t st);

// ---- dropout RNG contract (mirrors oracle/rng.py bit for bit) --------------------------------
__host__ __device__ __forceinline__ uint32_t cst_mix32(uint32_t seed, uint32_t stream, uint32_t idx) {
    uint32_t x = idx ^ (stream * 0x9E3779B1u);
    x = x * 0x85EBCA6Bu + seed;
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

struct CstDrop {          // p == 0 -> disabled
    float p;              // drop probability
    float scale;          // 1/(1-p)
    uint32_t thresh;      // floor(p * 2^24)
    uint32_t seed;        // host part of the seed
    uint32_t stream;      // call-site id
    const uint32_t* seed_dev;   // optional device word added to seed (graph replay)
    uint32_t base;        // added to every element index: (data-parallel rank) x (elements of this rank's mask tensor), see cst_set_drop_shard
};

// Data-parallel dropout contract: every mask tensor is batch-major, so element (global batch row, ...) of the one-process global
// batch has linear index rank * numel_local + local index on the rank that holds the row.  With the rank registered through
// cst_set_drop_shard() a shard draws exactly the masks the one-process run draws for its rows.  0 (default) = unsharded.
uint32_t cst_drop_shard_rank();

// numel = number of elements of the (local) tensor the mask applies to = the extent of the kernel's index space
static inline CstDrop cst_make_drop(float p, uint32_t seed, uint32_t stream, const void* seed_dev, long numel) {
    CstDrop d;
    d.p = p;
    d.scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    d.thresh = p > 0.f ? (uint32_t)floorf(p * 16777216.0f) : 0u;
    d.seed = seed;
    d.stream = stream;
    d.seed_dev = (const uint32_t*)seed_dev;
    d.base = p > 0.f ? cst_drop_shard_rank() * (uint32_t)numel : 0u;
    return d;
}

__device__ __forceinline__ uint32_t cst_drop_seed(const CstDrop& d) {
    return d.seed + (d.seed_dev ? *d.seed_dev : 0u);
}

__device__ __forceinline__ float cst_drop_mask(const CstDrop& d, uint32_t seed, uint32_t idx) {
    return ((cst_mix32(seed, d.stream, idx + d.base) >> 8) >= d.thresh) ? d.scale : 0.0f;
}

// ---- wave / block reductions -----------------------------------------------------------------
// On the DPP path: __shfl_xor lowers to ds_bpermute_b32, an LDS-pipe round trip per step (six dependent
// ones per wave reduction).  Inside each row of 16 lanes: swap inside pairs, swap pairs inside quads,
// mirror the half row, mirror the row (after the first two steps a quad holds one value, so the
// mirrors deliver the other quad's / the other half row's partial); the four row results are then read
// through SGPRs.  All 64 lanes must be active.
template <int CTRL>
__device__ __forceinline__ float cst_dpp(float v, float old) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += cst_dpp<0xB1>(v, 0.f);         // quad_perm [1,0,3,2]
    v += cst_dpp<0x4E>(v, 0.f);         // quad_perm [2,3,0,1]
    v += cst_dpp<0x141>(v, 0.f);        // row_half_mirror
    v += cst_dpp<0x140>(v, 0.f);        // row_mirror
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, cst_dpp<0xB1>(v, v));
    v = fmaxf(v, cst_dpp<0x4E>(v, v));
    v = fmaxf(v, cst_dpp<0x141>(v, v));
    v = fmaxf(v, cst_dpp<0x140>(v, v));
    return v;
}
__device__ __forceinline__ float cst_readlane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    return (cst_readlane(v, 0) + cst_readlane(v, 16)) + (cst_readlane(v, 32) + cst_readlane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    return fmaxf(fmaxf(cst_readlane(v, 0), cst_readlane(v, 16)), fmaxf(cst_readlane(v, 32), cst_readlane(v, 48)));
}

// block-wide reductions for blockDim.x multiple of 64, <= 1024.  `red` is >= 16 floats of LDS.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < nw; ++i) r = fmaxf(r, red[i]);
    return r;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

static inline int cst_div_up(long a, long b) { return (int)((a + b - 1) / b); }
