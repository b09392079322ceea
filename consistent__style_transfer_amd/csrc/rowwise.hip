// HBM-bound row kernels over (rows, V) or (rows, d) fp32 matrices:
//   * fused token cross-entropy forward+backward  (nn.CrossEntropyLoss at main_pretrain.py:73,
//     main_warmup.py:52, main_optimize.py:109,139 -- mean over rows, PAD rows included)
//   * temperature softmax + argmax over the vocabulary and its backward (rnn.py:83-85, 52-53)
//   * row argmax (rnn.py:92, main_optimize.py:104,131,162)
//   * residual + dropout + LayerNorm forward / backward (nn.TransformerEncoderLayer post-LN)
//   * column sums (bias gradients)
// One 256-thread block owns one vocabulary row and keeps it in registers between the reduction
// and the write, so a row is read from HBM once and written once (2 passes of algorithmic
// traffic).  Reductions are wavefront shuffles + one LDS hop.
#include "cst_common.h"

#define ROW_THREADS 256

// ---------------------------------------------------------------------------------------------
// row loader: NV4 float4 per thread when the row is 16-byte addressable, else scalar strided.
// ---------------------------------------------------------------------------------------------
template <int NV4, int NTH = ROW_THREADS>
struct RowRegs {
    float4 v[NV4];
    __device__ __forceinline__ void load(const float* __restrict__ row, int V, float fill) {
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c = (threadIdx.x + NTH * i) * 4;
            v[i] = (c < V) ? *reinterpret_cast<const float4*>(row + c) : make_float4(fill, fill, fill, fill);
        }
    }
    __device__ __forceinline__ void store(float* __restrict__ row, int V) const {
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c = (threadIdx.x + NTH * i) * 4;
            if (c < V) *reinterpret_cast<float4*>(row + c) = v[i];
        }
    }
};

__device__ __forceinline__ float f4max(float4 a) { return fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)); }

// ---------------------------------------------------------------------------------------------
// token CE, vector path
// ---------------------------------------------------------------------------------------------
template <int NV4>
__global__ __launch_bounds__(ROW_THREADS) void ce_vec_kernel(const float* __restrict__ logits, long ld,
                                                             const int64_t* __restrict__ target, int V,
                                                             float* __restrict__ row_loss,
                                                             float* dlogits, long ldd, float gscale,
                                                             unsigned short* __restrict__ dlb, long lddb) {
    __shared__ float red[16];
    const long r = blockIdx.x;
    const float* row = logits + r * ld;
    RowRegs<NV4> x;
    x.load(row, V, -INFINITY);
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV4; ++i) m = fmaxf(m, f4max(x.v[i]));
    m = block_max(m, red);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        x.v[i].x = expf(x.v[i].x - m); x.v[i].y = expf(x.v[i].y - m);
        x.v[i].z = expf(x.v[i].z - m); x.v[i].w = expf(x.v[i].w - m);
        s += (x.v[i].x + x.v[i].y) + (x.v[i].z + x.v[i].w);
    }
    s = block_sum(s, red);
    const long t = target[r];
    const bool valid = t >= 0 && t < V;
    if (threadIdx.x == 0) row_loss[r] = valid ? (logf(s) + m - row[t]) : 0.f;
    if (dlogits) {
        const float inv = gscale / s;
        __syncthreads();          // row[t] read above before an in-place overwrite
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c = (threadIdx.x + ROW_THREADS * i) * 4;
            float4 g = make_float4(x.v[i].x * inv, x.v[i].y * inv, x.v[i].z * inv, x.v[i].w * inv);
            if (valid && t >= c && t < c + 4) {
                if (t == c) g.x -= gscale; else if (t == c + 1) g.y -= gscale;
                else if (t == c + 2) g.z -= gscale; else g.w -= gscale;
            }
            x.v[i] = g;
            if (dlb && c < lddb) {                // bf16 twin, zero in the K padding: operand of the vocabulary dgrad / wgrad
                uint2 u = make_uint2(0u, 0u);
                if (c < V) {
                    __bf16 h0 = (__bf16)g.x, h1 = (__bf16)g.y, h2 = (__bf16)g.z, h3 = (__bf16)g.w;
                    u.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
                    u.y = (uint32_t)__builtin_bit_cast(unsigned short, h2) | ((uint32_t)__builtin_bit_cast(unsigned short, h3) << 16);
                }
                *reinterpret_cast<uint2*>(dlb + r * lddb + c) = u;
            }
        }
        x.store(dlogits + r * ldd, V);
    }
}

// generic path (any V / alignment): three strided sweeps, still one block per row
__global__ __launch_bounds__(ROW_THREADS) void ce_generic_kernel(const float* __restrict__ logits, long ld,
                                                                 const int64_t* __restrict__ target, int V,
                                                                 float* __restrict__ row_loss,
                                                                 float* dlogits, long ldd, float gscale) {
    __shared__ float red[16];
    const long r = blockIdx.x;
    const float* row = logits + r * ld;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < V; c += ROW_THREADS) m = fmaxf(m, row[c]);
    m = block_max(m, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += ROW_THREADS) s += expf(row[c] - m);
    s = block_sum(s, red);
    const long t = target[r];
    const bool valid = t >= 0 && t < V;
    if (threadIdx.x == 0) row_loss[r] = valid ? (logf(s) + m - row[t]) : 0.f;
    if (dlogits) {
        __syncthreads();
        const float inv = gscale / s;
        float* drow = dlogits + r * ldd;
        for (int c = threadIdx.x; c < V; c += ROW_THREADS) {
            float g = expf(row[c] - m) * inv;
            if (valid && c == t) g -= gscale;
            drow[c] = g;
        }
    }
}

static bool row_vec_ok(const void* p, long ld, int V) {
    return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0) && (V % 4 == 0) && V <= ROW_THREADS * 4 * 32;
}

#define ROW_DISPATCH(V, KERNEL, ...)                                                    \
    do {                                                                                \
        const int nv4 = cst_div_up(V, ROW_THREADS * 4);                                 \
        if (nv4 <= 1) hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__);                      \
        else if (nv4 <= 2) hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__);                 \
        else if (nv4 <= 4) hipLaunchKernelGGL((KERNEL<4>), __VA_ARGS__);                 \
        else if (nv4 <= 6) hipLaunchKernelGGL((KERNEL<6>), __VA_ARGS__);                 \
        else if (nv4 <= 10) hipLaunchKernelGGL((KERNEL<10>), __VA_ARGS__);               \
        else if (nv4 <= 16) hipLaunchKernelGGL((KERNEL<16>), __VA_ARGS__);               \
        else hipLaunchKernelGGL((KERNEL<32>), __VA_ARGS__);                              \
    } while (0)

extern "C" int cst_token_ce_b(const float* logits, long ld, const int64_t* target, int R, int V,
                              float* row_loss, float* dlogits, long ldd, float grad_scale,
                              void* dlogits_bf16, long lddb, void* stream) {
    CST_REQUIRE(logits && target && row_loss, "cst_token_ce: null pointer");
    CST_REQUIRE(R > 0 && V > 0 && ld >= V, "cst_token_ce: bad shape R=%d V=%d ld=%ld", R, V, ld);
    CST_REQUIRE(!dlogits || ldd >= V, "cst_token_ce: ldd too small");
    hipStream_t st = (hipStream_t)stream;
    unsigned short* dlb = (unsigned short*)dlogits_bf16;
    const bool vec = row_vec_ok(logits, ld, V) && (!dlogits || row_vec_ok(dlogits, ldd, V));
    CST_REQUIRE(!dlb || (dlogits && vec && V % 4 == 0 && lddb >= V && lddb % 4 == 0 && lddb <= ((V + 1023) / 1024) * 1024 && (((uintptr_t)dlb) & 7) == 0),
                "cst_token_ce: the bf16 gradient needs the vector path (V %% 4 == 0, 16-byte aligned rows) and V <= lddb <= V rounded up to 1024");
    if (vec) {
        ROW_DISPATCH(V, ce_vec_kernel, dim3(R), dim3(ROW_THREADS), 0, st, logits, ld, target, V, row_loss, dlogits, ldd, grad_scale, dlb, lddb);
    } else {
        hipLaunchKernelGGL(ce_generic_kernel, dim3(R), dim3(ROW_THREADS), 0, st, logits, ld, target, V, row_loss, dlogits, ldd, grad_scale);
    }
    CST_LAUNCH_CHECK("cst_token_ce");
    return CST_OK;
}

extern "C" int cst_token_ce(const float* logits, long ld, const int64_t* target, int R, int V,
                            float* row_loss, float* dlogits, long ldd, float grad_scale, void* stream) {
    return cst_token_ce_b(logits, ld, target, R, V, row_loss, dlogits, ldd, grad_scale, nullptr, 0, stream);
}

// ---------------------------------------------------------------------------------------------
// block argmax helper: first (lowest) index among equal maxima
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void block_argmax(float& bv, int& bi, float* redv, int* redi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { redv[w] = bv; redi[w] = bi; }
    __syncthreads();
    bv = redv[0]; bi = redi[0];
    for (int i = 1; i < nw; ++i)
        if (redv[i] > bv || (redv[i] == bv && redi[i] < bi)) { bv = redv[i]; bi = redi[i]; }
}

__device__ __forceinline__ void upd(float v, int i, float& bv, int& bi) {
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

// ---------------------------------------------------------------------------------------------
// Optional tail of the row kernels that know the row's argmax: the embedding of the token fed to the
// next decode step (rnn.py:88-95: scheduled sampling picks the fed-back id or the teacher token by a
// per-step coin; embedding dropout rnn.py:96), i.e. cst_embed_gather without its own launch.
// ---------------------------------------------------------------------------------------------
struct GatherTail {
    const float* table; long ldt;          // [V, E]; null = no tail
    float* out; long ldo; unsigned short* outb; long ldob;
    const int64_t* ids_b; long ldb; const int* coin;
    int E, V;
    CstDrop drop;
    unsigned short* pb; long ldpb; int wpb;   // softmax only: bf16 twin of the probabilities, zero in columns [V, wpb)
};

__device__ __forceinline__ void gather_tail(const GatherTail& t, long r, int id_a) {
    if (!t.table) return;
    long id = id_a;
    if (t.ids_b && !(t.coin && *t.coin)) id = t.ids_b[r * t.ldb];
    const uint32_t dseed = t.drop.p > 0.f ? cst_drop_seed(t.drop) : 0u;
    const bool ok = id >= 0 && id < t.V;
    const float* row = t.table + (ok ? id : 0) * t.ldt;
    for (int c = threadIdx.x; c < t.E; c += blockDim.x) {
        float v = ok ? row[c] : 0.f;
        if (t.drop.p > 0.f) v *= cst_drop_mask(t.drop, dseed, (uint32_t)(r * t.E + c));
        t.out[r * t.ldo + c] = v;
        if (t.outb) { __bf16 h = (__bf16)v; t.outb[r * t.ldob + c] = __builtin_bit_cast(unsigned short, h); }
    }
}

// ---------------------------------------------------------------------------------------------
// temperature softmax + argmax(p)
// ---------------------------------------------------------------------------------------------
template <int NV4, int NTH = ROW_THREADS>
__global__ __launch_bounds__(NTH) void softmax_tau_vec_kernel(const float* __restrict__ logits, long ld, float inv_tau,
                                                              float* __restrict__ p, long ldp,
                                                              int64_t* __restrict__ amax, int V, GatherTail tail) {
    __shared__ float red[16];
    __shared__ int redi[16];
    const long r = blockIdx.x;
    RowRegs<NV4, NTH> x;
    x.load(logits + r * ld, V, -INFINITY);
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        // the reference divides by tau first (rnn.py:83): softmax(logits / tau)
        x.v[i].x *= inv_tau; x.v[i].y *= inv_tau; x.v[i].z *= inv_tau; x.v[i].w *= inv_tau;
        m = fmaxf(m, f4max(x.v[i]));
    }
    m = block_max(m, red);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        x.v[i].x = expf(x.v[i].x - m); x.v[i].y = expf(x.v[i].y - m);
        x.v[i].z = expf(x.v[i].z - m); x.v[i].w = expf(x.v[i].w - m);
        s += (x.v[i].x + x.v[i].y) + (x.v[i].z + x.v[i].w);
    }
    s = block_sum(s, red);
    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int c = (threadIdx.x + NTH * i) * 4;
        x.v[i].x /= s; x.v[i].y /= s; x.v[i].z /= s; x.v[i].w /= s;
        if (c < V) { upd(x.v[i].x, c, bv, bi); upd(x.v[i].y, c + 1, bv, bi); upd(x.v[i].z, c + 2, bv, bi); upd(x.v[i].w, c + 3, bv, bi); }
    }
    x.store(p + r * ldp, V);
    if (tail.pb) {                                // operand of the soft-embedding products that consume all steps at once
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c = (threadIdx.x + NTH * i) * 4;
            if (c < tail.wpb) {
                uint2 u = make_uint2(0u, 0u);
                if (c < V) {
                    __bf16 h0 = (__bf16)x.v[i].x, h1 = (__bf16)x.v[i].y, h2 = (__bf16)x.v[i].z, h3 = (__bf16)x.v[i].w;
                    u.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
                    u.y = (uint32_t)__builtin_bit_cast(unsigned short, h2) | ((uint32_t)__builtin_bit_cast(unsigned short, h3) << 16);
                }
                *reinterpret_cast<uint2*>(tail.pb + r * tail.ldpb + c) = u;
            }
        }
    }
    if (amax || tail.table) {
        block_argmax(bv, bi, red, redi);
        if (amax && threadIdx.x == 0) amax[r] = bi;
        gather_tail(tail, r, bi);
    }
}

__global__ __launch_bounds__(ROW_THREADS) void softmax_tau_generic_kernel(const float* __restrict__ logits, long ld, float inv_tau,
                                                                          float* __restrict__ p, long ldp,
                                                                          int64_t* __restrict__ amax, int V, GatherTail tail) {
    __shared__ float red[16];
    __shared__ int redi[16];
    const long r = blockIdx.x;
    const float* row = logits + r * ld;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < V; c += ROW_THREADS) m = fmaxf(m, row[c] * inv_tau);
    m = block_max(m, red);
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += ROW_THREADS) s += expf(row[c] * inv_tau - m);
    s = block_sum(s, red);
    float bv = -INFINITY; int bi = 0x7fffffff;
    for (int c = threadIdx.x; c < V; c += ROW_THREADS) {
        const float q = expf(row[c] * inv_tau - m) / s;
        p[r * ldp + c] = q;
        upd(q, c, bv, bi);
    }
    if (amax || tail.table) {
        block_argmax(bv, bi, red, redi);
        if (amax && threadIdx.x == 0) amax[r] = bi;
        gather_tail(tail, r, bi);
    }
}

static GatherTail make_tail(const float* table, long ldt, int E, float* out, long ldo, void* out_bf16, long ldob,
                            const int64_t* ids_b, long ldb, const int* coin_dev, int V,
                            float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, int R) {
    GatherTail t;
    t.table = table; t.ldt = ldt; t.out = out; t.ldo = ldo; t.outb = (unsigned short*)out_bf16; t.ldob = ldob;
    t.ids_b = ids_b; t.ldb = ldb; t.coin = coin_dev; t.E = E; t.V = V;
    t.drop = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)R * E);      // mask index space: the gathered (R, E) rows
    t.pb = nullptr; t.ldpb = 0; t.wpb = 0;
    return t;
}

static int softmax_tau_launch(const float* logits, long ld, float inv_tau, float* p, long ldp,
                              int64_t* argmax_out, int R, int V, const GatherTail& tail, void* stream) {
    CST_REQUIRE(logits && p, "cst_softmax_tau: null pointer");
    CST_REQUIRE(R > 0 && V > 0 && ld >= V && ldp >= V, "cst_softmax_tau: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (tail.pb) {
        CST_REQUIRE(tail.wpb >= V && tail.wpb % 4 == 0 && tail.ldpb >= tail.wpb && tail.ldpb % 4 == 0 && (((uintptr_t)tail.pb) & 7) == 0,
                    "cst_softmax_tau_gather_b: bad bf16 output");
        CST_REQUIRE(row_vec_ok(logits, ld, V) && row_vec_ok(p, ldp, V) && tail.wpb < V + 64 && V <= 16384,
                    "cst_softmax_tau_gather_b: the bf16 twin needs the vector path (16-byte aligned rows, V %% 4 == 0, V <= 16384) and wpb < V + 64");
    }
    if (row_vec_ok(logits, ld, V) && row_vec_ok(p, ldp, V) && R <= 1024 && V > 4096 && V <= 12288) {
        // few rows (one decode step): 1024-thread workgroups keep 4x the loads in flight per CU
        hipLaunchKernelGGL((softmax_tau_vec_kernel<3, 1024>), dim3(R), dim3(1024), 0, st, logits, ld, inv_tau, p, ldp, argmax_out, V, tail);
    } else if (row_vec_ok(logits, ld, V) && row_vec_ok(p, ldp, V)) {
        ROW_DISPATCH(V, softmax_tau_vec_kernel, dim3(R), dim3(ROW_THREADS), 0, st, logits, ld, inv_tau, p, ldp, argmax_out, V, tail);
    } else {
        hipLaunchKernelGGL(softmax_tau_generic_kernel, dim3(R), dim3(ROW_THREADS), 0, st, logits, ld, inv_tau, p, ldp, argmax_out, V, tail);
    }
    CST_LAUNCH_CHECK("cst_softmax_tau");
    return CST_OK;
}

extern "C" int cst_softmax_tau(const float* logits, long ld, float inv_tau, float* p, long ldp,
                               int64_t* argmax_out, int R, int V, void* stream) {
    GatherTail none = make_tail(nullptr, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0.f, 0, 0, nullptr, 0);
    return softmax_tau_launch(logits, ld, inv_tau, p, ldp, argmax_out, R, V, none, stream);
}

extern "C" int cst_softmax_tau_gather(const float* logits, long ld, float inv_tau, float* p, long ldp,
                                      int64_t* argmax_out, int R, int V,
                                      const float* table, long ldt, int E, float* out, long ldo, void* out_bf16, long ldob,
                                      const int64_t* ids_b, long ldb, const int* coin_dev,
                                      float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                                      void* stream) {
    CST_REQUIRE(table && out && E > 0 && ldt >= E && ldo >= E, "cst_softmax_tau_gather: bad gather arguments");
    GatherTail t = make_tail(table, ldt, E, out, ldo, out_bf16, ldob, ids_b, ldb, coin_dev, V, drop_p, drop_seed, drop_stream, drop_seed_dev, R);
    return softmax_tau_launch(logits, ld, inv_tau, p, ldp, argmax_out, R, V, t, stream);
}

extern "C" int cst_softmax_tau_gather_b(const float* logits, long ld, float inv_tau, float* p, long ldp,
                                        void* p_bf16, long ldpb, int wpb, int64_t* argmax_out, int R, int V,
                                        const float* table, long ldt, int E, float* out, long ldo, void* out_bf16, long ldob,
                                        const int64_t* ids_b, long ldb, const int* coin_dev,
                                        float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                                        void* stream) {
    CST_REQUIRE(p_bf16, "cst_softmax_tau_gather_b: null bf16 output");
    CST_REQUIRE(!table || (out && E > 0 && ldt >= E && ldo >= E), "cst_softmax_tau_gather_b: bad gather arguments");
    GatherTail t = make_tail(table, ldt, E, out, ldo, out_bf16, ldob, ids_b, ldb, coin_dev, V, drop_p, drop_seed, drop_stream, drop_seed_dev, R);
    t.pb = (unsigned short*)p_bf16; t.ldpb = ldpb; t.wpb = wpb;
    return softmax_tau_launch(logits, ld, inv_tau, p, ldp, argmax_out, R, V, t, stream);
}

// backward: dx = inv_tau * p * (dp - sum(dp * p));  dx may alias dp
template <int NV4, int NTH = ROW_THREADS>
__global__ __launch_bounds__(NTH) void softmax_tau_bwd_vec_kernel(const float* __restrict__ p, long ldp,
                                                                  const float* dp, long lddp, float inv_tau,
                                                                  float* dx, long lddx, unsigned short* dxb, long lddxb, int V) {
    __shared__ float red[16];
    const long r = blockIdx.x;
    RowRegs<NV4, NTH> q, g;
    q.load(p + r * ldp, V, 0.f);
    g.load(dp + r * lddp, V, 0.f);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV4; ++i) s += (q.v[i].x * g.v[i].x + q.v[i].y * g.v[i].y) + (q.v[i].z * g.v[i].z + q.v[i].w * g.v[i].w);
    s = block_sum(s, red);
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        g.v[i].x = inv_tau * q.v[i].x * (g.v[i].x - s); g.v[i].y = inv_tau * q.v[i].y * (g.v[i].y - s);
        g.v[i].z = inv_tau * q.v[i].z * (g.v[i].z - s); g.v[i].w = inv_tau * q.v[i].w * (g.v[i].w - s);
    }
    g.store(dx + r * lddx, V);
    if (dxb) {                                    // bf16 copy: A operand of the fn_2 dgrad of this decode step
#pragma unroll
        for (int i = 0; i < NV4; ++i) {
            const int c = (threadIdx.x + NTH * i) * 4;
            if (c < V) {
                __bf16 h0 = (__bf16)g.v[i].x, h1 = (__bf16)g.v[i].y, h2 = (__bf16)g.v[i].z, h3 = (__bf16)g.v[i].w;
                uint2 u;
                u.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
                u.y = (uint32_t)__builtin_bit_cast(unsigned short, h2) | ((uint32_t)__builtin_bit_cast(unsigned short, h3) << 16);
                *reinterpret_cast<uint2*>(dxb + r * lddxb + c) = u;
            }
        }
    }
}

__global__ __launch_bounds__(ROW_THREADS) void softmax_tau_bwd_generic_kernel(const float* __restrict__ p, long ldp,
                                                                              const float* dp, long lddp, float inv_tau,
                                                                              float* dx, long lddx, unsigned short* dxb, long lddxb, int V) {
    __shared__ float red[16];
    const long r = blockIdx.x;
    float s = 0.f;
    for (int c = threadIdx.x; c < V; c += ROW_THREADS) s += p[r * ldp + c] * dp[r * lddp + c];
    s = block_sum(s, red);
    for (int c = threadIdx.x; c < V; c += ROW_THREADS) {
        const float v = inv_tau * p[r * ldp + c] * (dp[r * lddp + c] - s);
        dx[r * lddx + c] = v;
        if (dxb) { __bf16 h = (__bf16)v; dxb[r * lddxb + c] = __builtin_bit_cast(unsigned short, h); }
    }
}

extern "C" int cst_softmax_tau_bwd(const float* p, long ldp, const float* dp, long lddp, float inv_tau,
                                   float* dx, long lddx, void* dx_bf16, long lddxb, int R, int V, void* stream) {
    unsigned short* dxb = (unsigned short*)dx_bf16;
    CST_REQUIRE(!dxb || (lddxb >= V && lddxb % 4 == 0 && (((uintptr_t)dxb) & 7) == 0), "cst_softmax_tau_bwd: bad bf16 output");
    CST_REQUIRE(p && dp && dx, "cst_softmax_tau_bwd: null pointer");
    CST_REQUIRE(R > 0 && V > 0 && ldp >= V && lddp >= V && lddx >= V, "cst_softmax_tau_bwd: bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (row_vec_ok(p, ldp, V) && row_vec_ok(dp, lddp, V) && row_vec_ok(dx, lddx, V) && R <= 1024 && V > 4096 && V <= 12288) {
        hipLaunchKernelGGL((softmax_tau_bwd_vec_kernel<3, 1024>), dim3(R), dim3(1024), 0, st, p, ldp, dp, lddp, inv_tau, dx, lddx, dxb, lddxb, V);
    } else if (row_vec_ok(p, ldp, V) && row_vec_ok(dp, lddp, V) && row_vec_ok(dx, lddx, V) && V <= ROW_THREADS * 4 * 16) {
        const int nv4 = cst_div_up(V, ROW_THREADS * 4);
        if (nv4 <= 1) hipLaunchKernelGGL((softmax_tau_bwd_vec_kernel<1>), dim3(R), dim3(ROW_THREADS), 0, st, p, ldp, dp, lddp, inv_tau, dx, lddx, dxb, lddxb, V);
        else if (nv4 <= 4) hipLaunchKernelGGL((softmax_tau_bwd_vec_kernel<4>), dim3(R), dim3(ROW_THREADS), 0, st, p, ldp, dp, lddp, inv_tau, dx, lddx, dxb, lddxb, V);
        else if (nv4 <= 10) hipLaunchKernelGGL((softmax_tau_bwd_vec_kernel<10>), dim3(R), dim3(ROW_THREADS), 0, st, p, ldp, dp, lddp, inv_tau, dx, lddx, dxb, lddxb, V);
        else hipLaunchKernelGGL((softmax_tau_bwd_vec_kernel<16>), dim3(R), dim3(ROW_THREADS), 0, st, p, ldp, dp, lddp, inv_tau, dx, lddx, dxb, lddxb, V);
    } else {
        hipLaunchKernelGGL(softmax_tau_bwd_generic_kernel, dim3(R), dim3(ROW_THREADS), 0, st, p, ldp, dp, lddp, inv_tau, dx, lddx, dxb, lddxb, V);
    }
    CST_LAUNCH_CHECK("cst_softmax_tau_bwd");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// row argmax (first index among equal maxima), int64 out
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ x, long ld, int V, int vec,
                                                      int64_t* __restrict__ out, GatherTail tail) {
    __shared__ float red[16];
    __shared__ int redi[16];
    const long r = blockIdx.x;
    const float* row = x + r * ld;
    const int nth = blockDim.x;
    float bv = -INFINITY; int bi = 0x7fffffff;
    if (vec) {
        for (int c = threadIdx.x * 4; c < V; c += nth * 4) {
            const float4 t = *reinterpret_cast<const float4*>(row + c);
            upd(t.x, c, bv, bi); upd(t.y, c + 1, bv, bi); upd(t.z, c + 2, bv, bi); upd(t.w, c + 3, bv, bi);
        }
    } else {
        for (int c = threadIdx.x; c < V; c += nth) upd(row[c], c, bv, bi);
    }
    block_argmax(bv, bi, red, redi);
    if (bi == 0x7fffffff) bi = 0;
    if (threadIdx.x == 0) out[r] = bi;
    gather_tail(tail, r, bi);
}

static int argmax_launch(const float* x, long ld, int R, int V, int64_t* out, const GatherTail& tail, void* stream) {
    CST_REQUIRE(x && out && R > 0 && V > 0 && ld >= V, "cst_argmax_rows: bad arguments");
    const int vec = (((uintptr_t)x & 15) == 0) && (ld % 4 == 0) && (V % 4 == 0);
    const int nth = (R <= 1024 && V >= 4096) ? 1024 : ROW_THREADS;
    hipLaunchKernelGGL(argmax_kernel, dim3(R), dim3(nth), 0, (hipStream_t)stream, x, ld, V, vec, out, tail);
    CST_LAUNCH_CHECK("cst_argmax_rows");
    return CST_OK;
}

extern "C" int cst_argmax_rows(const float* x, long ld, int R, int V, int64_t* out, void* stream) {
    GatherTail none = make_tail(nullptr, 0, 0, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0.f, 0, 0, nullptr, 0);
    return argmax_launch(x, ld, R, V, out, none, stream);
}

extern "C" int cst_argmax_rows_gather(const float* x, long ld, int R, int V, int64_t* out,
                                      const float* table, long ldt, int E, float* gout, long ldo, void* out_bf16, long ldob,
                                      const int64_t* ids_b, long ldb, const int* coin_dev,
                                      float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                                      void* stream) {
    CST_REQUIRE(table && gout && E > 0 && ldt >= E && ldo >= E, "cst_argmax_rows_gather: bad gather arguments");
    GatherTail t = make_tail(table, ldt, E, gout, ldo, out_bf16, ldob, ids_b, ldb, coin_dev, V, drop_p, drop_seed, drop_stream, drop_seed_dev, R);
    return argmax_launch(x, ld, R, V, out, t, stream);
}

// ---------------------------------------------------------------------------------------------
// z = res + dropout(x) ; y = LayerNorm(z) * gamma + beta.   One wavefront per row of d <= 1024.
// ---------------------------------------------------------------------------------------------
#define LN_MAXE 16
template <int NE>
__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                float eps, float* z, float* __restrict__ y,
                                                                float* __restrict__ mean, float* __restrict__ rstd,
                                                                int T, int d, CstDrop drop,
                                                                unsigned short* __restrict__ yb, long ldyb) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    float v[NE];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + 64 * i;
        float t = 0.f;
        if (c < d) {
            t = x[row * d + c];
            if (drop.p > 0.f) t *= cst_drop_mask(drop, dseed, (uint32_t)(row * d + c));
            if (res) t += res[row * d + c];
        }
        v[i] = t;
        s += t;
    }
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + 64 * i;
        const float t = (c < d) ? v[i] - mu : 0.f;
        q += t * t;
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int c = lane + 64 * i;
        if (c < d) {
            if (z) z[row * d + c] = v[i];
            const float yv = (v[i] - mu) * rs * gamma[c] + beta[c];
            y[row * d + c] = yv;
            if (yb) { __bf16 h = (__bf16)yv; yb[row * ldyb + c] = __builtin_bit_cast(unsigned short, h); }   // next GEMM's A operand
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

// d == 256 * NV and 16-byte aligned operands (the module constants 512 / 768 / 1024): one float4 per lane per 256 columns,
// every load of the row in flight before the first use, 16-byte stores (8-byte for the bf16 twin)
template <int NV>
__global__ __launch_bounds__(256) void add_layernorm_fwd_vec_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                    float eps, float* z, float* __restrict__ y,
                                                                    float* __restrict__ mean, float* __restrict__ rstd,
                                                                    int T, CstDrop drop, unsigned short* __restrict__ yb, long ldyb) {
    constexpr int d = 256 * NV;
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= T) return;
    float4 v[NV], r[NV], ga[NV], be[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const float4*>(x + row * d + 4 * lane + 256 * i);
    if (res) {
#pragma unroll
        for (int i = 0; i < NV; ++i) r[i] = *reinterpret_cast<const float4*>(res + row * d + 4 * lane + 256 * i);
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ga[i] = *reinterpret_cast<const float4*>(gamma + 4 * lane + 256 * i);
        be[i] = *reinterpret_cast<const float4*>(beta + 4 * lane + 256 * i);
    }
    if (drop.p > 0.f) {
        const uint32_t dseed = cst_drop_seed(drop);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const uint32_t e = (uint32_t)(row * d + 4 * lane + 256 * i);
            v[i].x *= cst_drop_mask(drop, dseed, e); v[i].y *= cst_drop_mask(drop, dseed, e + 1);
            v[i].z *= cst_drop_mask(drop, dseed, e + 2); v[i].w *= cst_drop_mask(drop, dseed, e + 3);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i].x += r[i].x; v[i].y += r[i].y; v[i].z += r[i].z; v[i].w += r[i].w;
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mu = wave_sum(s) / (float)d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mu, b = v[i].y - mu, c = v[i].z - mu, e = v[i].w - mu;
        q += (a * a + b * b) + (c * c + e * e);
    }
    const float rs = 1.0f / sqrtf(wave_sum(q) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const long o = row * d + 4 * lane + 256 * i;
        if (z) *reinterpret_cast<float4*>(z + o) = v[i];
        float4 yv;
        yv.x = (v[i].x - mu) * rs * ga[i].x + be[i].x; yv.y = (v[i].y - mu) * rs * ga[i].y + be[i].y;
        yv.z = (v[i].z - mu) * rs * ga[i].z + be[i].z; yv.w = (v[i].w - mu) * rs * ga[i].w + be[i].w;
        *reinterpret_cast<float4*>(y + o) = yv;
        if (yb) {                                                     // next GEMM's A operand
            __bf16 h0 = (__bf16)yv.x, h1 = (__bf16)yv.y, h2 = (__bf16)yv.z, h3 = (__bf16)yv.w;
            uint2 u;
            u.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
            u.y = (uint32_t)__builtin_bit_cast(unsigned short, h2) | ((uint32_t)__builtin_bit_cast(unsigned short, h3) << 16);
            *reinterpret_cast<uint2*>(yb + row * ldyb + 4 * lane + 256 * i) = u;
        }
    }
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
}

extern "C" int cst_add_layernorm_fwd_b(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                                       float* z, float* y, float* mean, float* rstd, int T, int d,
                                       float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                                       void* y_bf16, long ldyb, void* stream) {
    CST_REQUIRE(x && gamma && beta && y && mean && rstd, "cst_add_layernorm_fwd: null pointer");
    CST_REQUIRE(T > 0 && d > 0 && d <= 64 * LN_MAXE, "cst_add_layernorm_fwd: d=%d unsupported (max %d)", d, 64 * LN_MAXE);
    CST_REQUIRE(!y_bf16 || ldyb >= d, "cst_add_layernorm_fwd: bf16 leading dimension < d");
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)T * d);
    const uintptr_t al = (uintptr_t)x | (uintptr_t)res | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)z | (uintptr_t)y;
    if (d % 256 == 0 && d <= 1024 && (al & 15) == 0 && (!y_bf16 || (ldyb % 4 == 0 && (((uintptr_t)y_bf16) & 7) == 0))) {
#define LN_FWD_V(NV) hipLaunchKernelGGL(add_layernorm_fwd_vec_kernel<NV>, dim3(cst_div_up(T, 4)), dim3(256), 0, (hipStream_t)stream, \
                                        x, res, gamma, beta, eps, z, y, mean, rstd, T, dr, (unsigned short*)y_bf16, ldyb)
        if (d == 256) LN_FWD_V(1); else if (d == 512) LN_FWD_V(2); else if (d == 768) LN_FWD_V(3); else LN_FWD_V(4);
#undef LN_FWD_V
        CST_LAUNCH_CHECK("cst_add_layernorm_fwd");
        return CST_OK;
    }
#define LN_FWD(NEV) hipLaunchKernelGGL(add_layernorm_fwd_kernel<NEV>, dim3(cst_div_up(T, 4)), dim3(256), 0, (hipStream_t)stream, \
                                       x, res, gamma, beta, eps, z, y, mean, rstd, T, d, dr, (unsigned short*)y_bf16, ldyb)
    const int ne = cst_div_up(d, 64);
    if (ne <= 4) LN_FWD(4); else if (ne <= 8) LN_FWD(8); else if (ne <= 12) LN_FWD(12); else LN_FWD(16);
#undef LN_FWD
    CST_LAUNCH_CHECK("cst_add_layernorm_fwd");
    return CST_OK;
}

extern "C" int cst_add_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                                     float* z, float* y, float* mean, float* rstd, int T, int d,
                                     float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                                     void* stream) {
    return cst_add_layernorm_fwd_b(x, res, gamma, beta, eps, z, y, mean, rstd, T, d, drop_p, drop_seed, drop_stream, drop_seed_dev,
                                   nullptr, 0, stream);
}

// backward: dz = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma;
// per-block partial column sums of dy*xhat (dgamma) and dy (dbeta) go to part[2][nblk][d].
template <int NE>      // NE = ceil(d / 64) column groups per lane: arrays sized for the actual width, not for the 1024 maximum
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            const float* __restrict__ gamma, float* dz,
                                                            float* __restrict__ part, int T, int d, int rows_per_block,
                                                            unsigned short* __restrict__ dzb, long lddzb, CstDrop bdrop, int three) {
    // three != 0: a third partial, the column sums of dropout'(dz) (the bias gradient of the Linear in front of this
    // LayerNorm), and the partials laid out [block][3][d] so that ONE column-sum launch finishes all three
    __shared__ float sh[3][4][256];      // only used for d <= 256 per pass; see loop below
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float ag[NE], ab[NE], az[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) { ag[i] = 0.f; ab[i] = 0.f; az[i] = 0.f; }
    const long r0 = (long)blockIdx.x * rows_per_block;
    const uint32_t bseed = (dzb && bdrop.p > 0.f) ? cst_drop_seed(bdrop) : 0u;
    for (long row = r0 + w; row < r0 + rows_per_block && row < T; row += 4) {
        const float mu = mean[row], rs = rstd[row];
        float xh[NE], g[NE];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + 64 * i;
            if (c < d) {
                const float dyv = dy[row * d + c];
                xh[i] = (z[row * d + c] - mu) * rs;
                g[i] = dyv * gamma[c];
                s1 += g[i]; s2 += g[i] * xh[i];
                ag[i] += dyv * xh[i]; ab[i] += dyv;
            } else { xh[i] = 0.f; g[i] = 0.f; }
        }
        s1 = wave_sum(s1) / (float)d; s2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int c = lane + 64 * i;
            if (c < d) {
                const float dv = rs * (g[i] - s1 - xh[i] * s2);
                dz[row * d + c] = dv;
                if (dzb) {          // dropout'(dz) in bf16: A operand of the dgrad / weight-gradient GEMMs behind this LayerNorm
                    float t = dv;
                    if (bdrop.p > 0.f) t *= cst_drop_mask(bdrop, bseed, (uint32_t)(row * d + c));
                    __bf16 h = (__bf16)t;
                    dzb[row * lddzb + c] = __builtin_bit_cast(unsigned short, h);
                    az[i] += t;
                }
            }
        }
    }
    // reduce the 4 waves' partial sums through LDS, 4 column groups (of 64) at a time
    const long nblk = gridDim.x;
#pragma unroll
    for (int i0 = 0; i0 < NE; i0 += 4) {
        if (i0 * 64 >= d) break;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i0 + i < NE) {
                sh[0][w][i * 64 + lane] = ag[i0 + i]; sh[1][w][i * 64 + lane] = ab[i0 + i]; sh[2][w][i * 64 + lane] = az[i0 + i];
            }
        }
        __syncthreads();
        // 256 threads <-> 256 columns of this pass
        const int c = i0 * 64 + threadIdx.x;
        if (c < d) {
            const float sg = (sh[0][0][threadIdx.x] + sh[0][1][threadIdx.x]) + (sh[0][2][threadIdx.x] + sh[0][3][threadIdx.x]);
            const float sb = (sh[1][0][threadIdx.x] + sh[1][1][threadIdx.x]) + (sh[1][2][threadIdx.x] + sh[1][3][threadIdx.x]);
            if (three) {
                const float sz = (sh[2][0][threadIdx.x] + sh[2][1][threadIdx.x]) + (sh[2][2][threadIdx.x] + sh[2][3][threadIdx.x]);
                float* pr = part + (long)blockIdx.x * 3 * d;
                pr[c] = sg; pr[d + c] = sb; pr[2 * d + c] = sz;
            } else {
                part[(0 * nblk + blockIdx.x) * d + c] = sg;
                part[(1 * nblk + blockIdx.x) * d + c] = sb;
            }
        }
    }
}

// d == 256 * NV, 16-byte aligned operands: float4 per lane per 256 columns, the next row of the wave requested before the
// current one is reduced (the per-row chain load -> two wave reductions -> store otherwise leaves the memory pipe idle).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_bwd_vec_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, float* dz,
                                                                float* __restrict__ part, int T, int rows_per_block,
                                                                unsigned short* __restrict__ dzb, long lddzb, CstDrop bdrop, int three) {
    constexpr int d = 256 * NV;
    __shared__ float sh[3][4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float4 ag[NV], ab[NV], az[NV], ga[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        ag[i] = ab[i] = az[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        ga[i] = *reinterpret_cast<const float4*>(gamma + 4 * lane + 256 * i);
    }
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long rend = min((long)T, r0 + rows_per_block);
    const uint32_t bseed = (dzb && bdrop.p > 0.f) ? cst_drop_seed(bdrop) : 0u;
    float4 dyn[NV], zn[NV];
    float mun = 0.f, rsn = 0.f;
    long row = r0 + w;
    if (row < rend) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            dyn[i] = *reinterpret_cast<const float4*>(dy + row * d + 4 * lane + 256 * i);
            zn[i] = *reinterpret_cast<const float4*>(z + row * d + 4 * lane + 256 * i);
        }
        mun = mean[row]; rsn = rstd[row];
    }
    while (row < rend) {
        float4 dyv[NV], xh[NV], g[NV];
        const float mu = mun, rs = rsn;
#pragma unroll
        for (int i = 0; i < NV; ++i) { dyv[i] = dyn[i]; xh[i] = zn[i]; }
        const long nrow = row + 4;
        if (nrow < rend) {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                dyn[i] = *reinterpret_cast<const float4*>(dy + nrow * d + 4 * lane + 256 * i);
                zn[i] = *reinterpret_cast<const float4*>(z + nrow * d + 4 * lane + 256 * i);
            }
            mun = mean[nrow]; rsn = rstd[nrow];
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            xh[i].x = (xh[i].x - mu) * rs; xh[i].y = (xh[i].y - mu) * rs; xh[i].z = (xh[i].z - mu) * rs; xh[i].w = (xh[i].w - mu) * rs;
            g[i].x = dyv[i].x * ga[i].x; g[i].y = dyv[i].y * ga[i].y; g[i].z = dyv[i].z * ga[i].z; g[i].w = dyv[i].w * ga[i].w;
            s1 += (g[i].x + g[i].y) + (g[i].z + g[i].w);
            s2 += (g[i].x * xh[i].x + g[i].y * xh[i].y) + (g[i].z * xh[i].z + g[i].w * xh[i].w);
            ag[i].x += dyv[i].x * xh[i].x; ag[i].y += dyv[i].y * xh[i].y; ag[i].z += dyv[i].z * xh[i].z; ag[i].w += dyv[i].w * xh[i].w;
            ab[i].x += dyv[i].x; ab[i].y += dyv[i].y; ab[i].z += dyv[i].z; ab[i].w += dyv[i].w;
        }
        s1 = wave_sum(s1) / (float)d; s2 = wave_sum(s2) / (float)d;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const long o = row * d + 4 * lane + 256 * i;
            float4 dv;
            dv.x = rs * (g[i].x - s1 - xh[i].x * s2); dv.y = rs * (g[i].y - s1 - xh[i].y * s2);
            dv.z = rs * (g[i].z - s1 - xh[i].z * s2); dv.w = rs * (g[i].w - s1 - xh[i].w * s2);
            *reinterpret_cast<float4*>(dz + o) = dv;
            if (dzb) {          // dropout'(dz) in bf16: A operand of the dgrad / weight-gradient GEMMs behind this LayerNorm
                float4 t = dv;
                if (bdrop.p > 0.f) {
                    const uint32_t e = (uint32_t)o;
                    t.x *= cst_drop_mask(bdrop, bseed, e); t.y *= cst_drop_mask(bdrop, bseed, e + 1);
                    t.z *= cst_drop_mask(bdrop, bseed, e + 2); t.w *= cst_drop_mask(bdrop, bseed, e + 3);
                }
                __bf16 h0 = (__bf16)t.x, h1 = (__bf16)t.y, h2 = (__bf16)t.z, h3 = (__bf16)t.w;
                uint2 u;
                u.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
                u.y = (uint32_t)__builtin_bit_cast(unsigned short, h2) | ((uint32_t)__builtin_bit_cast(unsigned short, h3) << 16);
                *reinterpret_cast<uint2*>(dzb + row * lddzb + 4 * lane + 256 * i) = u;
                az[i].x += t.x; az[i].y += t.y; az[i].z += t.z; az[i].w += t.w;
            }
        }
        row = nrow;
    }
    // the 4 waves' partial column sums through LDS, 256 columns at a time
    const long nblk = gridDim.x;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        __syncthreads();
        *reinterpret_cast<float4*>(&sh[0][w][4 * lane]) = ag[i];
        *reinterpret_cast<float4*>(&sh[1][w][4 * lane]) = ab[i];
        *reinterpret_cast<float4*>(&sh[2][w][4 * lane]) = az[i];
        __syncthreads();
        const int t = threadIdx.x, c = 256 * i + t;
        const float sg = (sh[0][0][t] + sh[0][1][t]) + (sh[0][2][t] + sh[0][3][t]);
        const float sb = (sh[1][0][t] + sh[1][1][t]) + (sh[1][2][t] + sh[1][3][t]);
        if (three) {
            const float sz = (sh[2][0][t] + sh[2][1][t]) + (sh[2][2][t] + sh[2][3][t]);
            float* pr = part + (long)blockIdx.x * 3 * d;
            pr[c] = sg; pr[d + c] = sb; pr[2 * d + c] = sz;
        } else {
            part[(0 * nblk + blockIdx.x) * d + c] = sg;
            part[(1 * nblk + blockIdx.x) * d + c] = sb;
        }
    }
}

// out[c] (+)= sum_r X[r, c]   (r over M rows); grid (colgroups, rowsplits); atomics when split
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ X, long ld, int M, int N,
                                                     float* __restrict__ out, int rows_per_split, int use_atomic, int accumulate) {
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const long r0 = (long)blockIdx.y * rows_per_split;
    float s = 0.f;
    if (c < N) {
        const long rend = min((long)M, r0 + rows_per_split);
        long r = r0 + w;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;       // four loads in flight per lane
        for (; r + 12 < rend; r += 16) {
            s0 += X[r * ld + c]; s1 += X[(r + 4) * ld + c]; s2 += X[(r + 8) * ld + c]; s3 += X[(r + 12) * ld + c];
        }
        for (; r < rend; r += 4) s0 += X[r * ld + c];
        s = (s0 + s1) + (s2 + s3);
    }
    sh[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < N) {
        const float t = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
        if (use_atomic) atomicAdd(out + c, t);
        else out[c] = accumulate ? out[c] + t : t;
    }
}

extern "C" int cst_colsum(const float* X, long ld, int M, int N, float* out, int accumulate, void* stream) {
    CST_REQUIRE(X && out && M > 0 && N > 0 && ld >= N, "cst_colsum: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int cg = cst_div_up(N, 64);
    int splits = 1;
    if (M >= 256) { splits = 2048 / cg; if (splits < 1) splits = 1; if (splits > M / 32) splits = M / 32; if (splits < 1) splits = 1; }
    const int rps = cst_div_up(M, splits);
    const int use_atomic = splits > 1;
    if (use_atomic && !accumulate) {
        if (cst_zero_words(out, N, st) != CST_OK) { cst_set_error("cst_colsum: zero fill failed"); return CST_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(colsum_kernel, dim3(cg, splits), dim3(256), 0, st, X, ld, M, N, out, rps, use_atomic, accumulate);
    CST_LAUNCH_CHECK("cst_colsum");
    return CST_OK;
}

extern "C" long cst_layernorm_bwd_workspace_floats(int T, int d) {
    const int nblk = T < 4096 ? cst_div_up(T, 4) : 1024;
    return 2L * nblk * d;
}

extern "C" int cst_layernorm_bwd_b(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                                   float* dz, float* dgamma, float* dbeta, int accumulate,
                                   float* workspace, long workspace_floats, int T, int d,
                                   void* dz_bf16, long lddzb, float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                                   float* dparams3, void* stream) {
    CST_REQUIRE(dy && z && mean && rstd && gamma && dz && workspace, "cst_layernorm_bwd: null pointer");
    CST_REQUIRE(T > 0 && d > 0 && d <= 64 * LN_MAXE, "cst_layernorm_bwd: d=%d unsupported", d);
    CST_REQUIRE(!dz_bf16 || lddzb >= d, "cst_layernorm_bwd: bf16 leading dimension < d");
    CST_REQUIRE(!dparams3 || dz_bf16, "cst_layernorm_bwd: the fused bias gradient needs the bf16 output");
    const int nblk = T < 4096 ? cst_div_up(T, 4) : 1024;      // one row per wave per pass: latency hidden by occupancy
    const long need = (dparams3 ? 3L : 2L) * nblk * d;
    CST_REQUIRE(workspace_floats >= need, "cst_layernorm_bwd: workspace too small (%ld < %ld)", workspace_floats, need);
    const int rpb = cst_div_up(T, nblk);
    hipStream_t st = (hipStream_t)stream;
    CstDrop bd = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)T * d);
    const uintptr_t al = (uintptr_t)dy | (uintptr_t)z | (uintptr_t)gamma | (uintptr_t)dz;
    const bool vec = d % 256 == 0 && d <= 1024 && (al & 15) == 0 && (!dz_bf16 || (lddzb % 4 == 0 && (((uintptr_t)dz_bf16) & 7) == 0));
#define LN_BWD_V(NV) hipLaunchKernelGGL(layernorm_bwd_vec_kernel<NV>, dim3(nblk), dim3(256), 0, st, dy, z, mean, rstd, gamma, dz, workspace, T, rpb, \
                                        (unsigned short*)dz_bf16, lddzb, bd, dparams3 ? 1 : 0)
#define LN_BWD(NEV) hipLaunchKernelGGL(layernorm_bwd_kernel<NEV>, dim3(nblk), dim3(256), 0, st, dy, z, mean, rstd, gamma, dz, workspace, T, d, rpb, \
                                       (unsigned short*)dz_bf16, lddzb, bd, dparams3 ? 1 : 0)
    const int ne = cst_div_up(d, 64);
    if (vec) { if (d == 256) LN_BWD_V(1); else if (d == 512) LN_BWD_V(2); else if (d == 768) LN_BWD_V(3); else LN_BWD_V(4); }
    else if (ne <= 4) LN_BWD(4); else if (ne <= 8) LN_BWD(8); else if (ne <= 12) LN_BWD(12); else LN_BWD(16);
#undef LN_BWD
#undef LN_BWD_V
    CST_LAUNCH_CHECK("cst_layernorm_bwd");
    if (dparams3)                                         // (dgamma | dbeta | dbias) in one pass over [nblk, 3d]
        return cst_colsum(workspace, 3L * d, nblk, 3 * d, dparams3, accumulate, stream);
    if (dgamma) {
        int rc = cst_colsum(workspace, d, nblk, d, dgamma, accumulate, stream);
        if (rc) return rc;
    }
    if (dbeta) {
        int rc = cst_colsum(workspace + (long)nblk * d, d, nblk, d, dbeta, accumulate, stream);
        if (rc) return rc;
    }
    return CST_OK;
}

extern "C" int cst_layernorm_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                                 float* dz, float* dgamma, float* dbeta, int accumulate,
                                 float* workspace, long workspace_floats, int T, int d, void* stream) {
    return cst_layernorm_bwd_b(dy, z, mean, rstd, gamma, dz, dgamma, dbeta, accumulate, workspace, workspace_floats, T, d,
                               nullptr, 0, 0.f, 0, 0, nullptr, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------
// out[0] = scale * sum(in[0..n))   (deterministic single-block reduction of per-row losses)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void reduce_sum_kernel(const float* __restrict__ in, long n, float scale, float* __restrict__ out, int accumulate) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) s += in[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s * scale : s * scale;
}

extern "C" int cst_reduce_sum(const float* in, long n, float scale, float* out, int accumulate, void* stream) {
    CST_REQUIRE(in && out && n > 0, "cst_reduce_sum: bad arguments");
    hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, in, n, scale, out, accumulate);
    CST_LAUNCH_CHECK("cst_reduce_sum");
    return CST_OK;
}
