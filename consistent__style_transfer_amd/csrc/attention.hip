// Attention kernels for gfx950 (exact fp32 VALU; tiles are tiny: S <= 64 keys).
//
//  (1) Unmasked multi-head self-attention core of nn.TransformerEncoderLayer as the reference
//      configures it (mlm.py:20-22, match.py:18-20; no mask is ever passed, mlm.py:43,
//      match.py:39): per (batch row, head), P = softmax(Q K^T / sqrt(hd)), attention dropout on
//      P, O = P V.  S = L (MLM) or L1+L2 (Matcher) <= 64.
//      One 256-thread workgroup per (b, h); its 4 wavefronts share the head's K/V.  In the
//      forward each lane keeps one key row (scores) and one value column (output) in registers
//      and a wave exchanges the probability row through LDS; softmax reductions are wavefront
//      shuffles.  The backward stages Q, K, V, dO and the S x S matrices P and dS in LDS.
//  (2) Single-query dot-product attention of the generator's decoder (rnn.py:46-50, called at
//      rnn.py:76): one workgroup per batch row, memory (L' <= 64 rows of D) streamed once.
#include "cst_common.h"

#define MHA_SMAX 64

// ---------------------------------------------------------------------------------------------
// MHA forward.  qkv [B,S,3d] (q | k | v thirds, heads interleaved inside each third as torch's
// packed in_proj does), out [B,S,d], lse [B,H,S].
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void mha_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                      float* __restrict__ lse, int S, int H, float scale, CstDrop drop) {
    constexpr int HD4 = HD / 4;
    constexpr int NC = (HD + 63) / 64;                 // output columns per lane
    __shared__ __attribute__((aligned(16))) float Qs[MHA_SMAX * HD];
    __shared__ __attribute__((aligned(16))) float Ps[4][MHA_SMAX];
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int d = H * HD;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* base = qkv + (long)b * S * 3 * d + h * HD;

    // Q tile -> LDS (coalesced over the head dimension)
    for (int e = threadIdx.x; e < S * HD; e += 256) {
        const int i = e / HD, c = e % HD;
        Qs[i * HD + c] = base[(long)i * 3 * d + c];
    }
    // key row `lane` -> registers
    float kreg[HD];
    if (lane < S) {
        const float* kp = base + (long)lane * 3 * d + d;
#pragma unroll
        for (int c = 0; c < HD; ++c) kreg[c] = kp[c];
    } else {
#pragma unroll
        for (int c = 0; c < HD; ++c) kreg[c] = 0.f;
    }
    // value columns lane (+64) -> registers, all S keys
    float vreg[NC][MHA_SMAX];
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
        const int c = lane + 64 * cc;
#pragma unroll
        for (int j = 0; j < MHA_SMAX; ++j)
            vreg[cc][j] = (j < S && c < HD) ? base[(long)j * 3 * d + 2 * d + c] : 0.f;
    }
    __syncthreads();
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;

    for (int i = w; i < S; i += 4) {
        float s = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < HD4; ++c4) {
            const float4 q = *reinterpret_cast<const float4*>(&Qs[i * HD + c4 * 4]);
            s += q.x * kreg[c4 * 4 + 0] + q.y * kreg[c4 * 4 + 1] + q.z * kreg[c4 * 4 + 2] + q.w * kreg[c4 * 4 + 3];
        }
        s = (lane < S) ? s * scale : -INFINITY;
        const float m = wave_max(s);
        const float e = (lane < S) ? expf(s - m) : 0.f;
        const float sum = wave_sum(e);
        float p = e / sum;
        if (lane == 0) lse[((long)b * H + h) * S + i] = m + logf(sum);
        if (drop.p > 0.f && lane < S)
            p *= cst_drop_mask(drop, dseed, (uint32_t)((((long)b * H + h) * S + i) * S + lane));
        Ps[w][lane] = p;
        __builtin_amdgcn_wave_barrier();
        // O[i][c] = sum_j P[i][j] V[j][c]; this wave's own LDS row, lanes now index columns
        float o[NC];
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) o[cc] = 0.f;
#pragma unroll
        for (int j4 = 0; j4 < MHA_SMAX / 4; ++j4) {
            if (j4 * 4 < S) {
                const float4 pj = *reinterpret_cast<const float4*>(&Ps[w][j4 * 4]);
#pragma unroll
                for (int cc = 0; cc < NC; ++cc)
                    o[cc] += pj.x * vreg[cc][j4 * 4 + 0] + pj.y * vreg[cc][j4 * 4 + 1] + pj.z * vreg[cc][j4 * 4 + 2] + pj.w * vreg[cc][j4 * 4 + 3];
            }
        }
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
            const int c = lane + 64 * cc;
            if (c < HD) out[((long)b * S + i) * d + h * HD + c] = o[cc];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int cst_mha_fwd(const float* qkv, float* out, float* lse, int B, int S, int H, int hd,
                           float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                           void* stream) {
    CST_REQUIRE(qkv && out && lse, "cst_mha_fwd: null pointer");
    CST_REQUIRE(B > 0 && S > 0 && S <= MHA_SMAX && H > 0, "cst_mha_fwd: S=%d unsupported (max %d)", S, MHA_SMAX);
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev);
    const float scale = 1.0f / sqrtf((float)hd);
    dim3 grid(B * H), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (hd) {
        case 8: hipLaunchKernelGGL((mha_fwd_kernel<8>), grid, block, 0, st, qkv, out, lse, S, H, scale, dr); break;
        case 16: hipLaunchKernelGGL((mha_fwd_kernel<16>), grid, block, 0, st, qkv, out, lse, S, H, scale, dr); break;
        case 32: hipLaunchKernelGGL((mha_fwd_kernel<32>), grid, block, 0, st, qkv, out, lse, S, H, scale, dr); break;
        case 64: hipLaunchKernelGGL((mha_fwd_kernel<64>), grid, block, 0, st, qkv, out, lse, S, H, scale, dr); break;
        case 96: hipLaunchKernelGGL((mha_fwd_kernel<96>), grid, block, 0, st, qkv, out, lse, S, H, scale, dr); break;
        default: cst_set_error("cst_mha_fwd: head dim %d unsupported (8, 16, 32, 64, 96)", hd); return CST_ERR_ARG;
    }
    CST_LAUNCH_CHECK("cst_mha_fwd");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// MHA backward: recompute P from Q, K and the saved log-sum-exp, then
//   dV = Pd^T dO ; dPd = dO V^T ; dS = P o (dP - rowsum(dP o P)) * scale ; dQ = dS K ; dK = dS^T Q
// with Pd the dropped probabilities.  dqkv [B,S,3d].
// ---------------------------------------------------------------------------------------------
template <int HD>
__global__ __launch_bounds__(256) void mha_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                      const float* __restrict__ lse, float* __restrict__ dqkv,
                                                      int S, int H, float scale, CstDrop drop) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HD4 = HD / 4;
    constexpr int RMAX = MHA_SMAX / 4;   // rows per wave (row i belongs to wave i % 4)
    const int SP = (S + 3) / 4 * 4 + 4;  // row stride of the S x S images: multiple of 4 (b128 reads), keys >= S hold 0
    float* Qs = smem;                    // [S][HD]
    float* Ks = Qs + S * HD;             // [S][HD]
    float* Vs = Ks + S * HD;             // [S][HD]
    float* Os = Vs + S * HD;             // [S][HD]  (dO)
    float* Dm = Os + S * HD;             // [S][SP]  dS[i][j]
    float* DmT = Dm + S * SP;            // [S][SP]  dS[j][i]
    float* PmT = DmT + S * SP;           // [S][SP]  Pd[j][i] (dropped probabilities, transposed)
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int d = H * HD;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* base = qkv + (long)b * S * 3 * d + h * HD;
    const float* dob = dout + (long)b * S * d + h * HD;
    for (int e = threadIdx.x; e < S * HD; e += 256) {
        const int i = e / HD, c = e % HD;
        Qs[e] = base[(long)i * 3 * d + c];
        Ks[e] = base[(long)i * 3 * d + d + c];
        Vs[e] = base[(long)i * 3 * d + 2 * d + c];
        Os[e] = dob[(long)i * d + c];
    }
    for (int e = threadIdx.x; e < 3 * S * SP; e += 256) Dm[e] = 0.f;      // zero incl. the padding keys
    __syncthreads();
    // pass 1: lane = key j; key and value rows in registers
    float kreg[HD], vreg[HD];
#pragma unroll
    for (int c = 0; c < HD; ++c) {
        kreg[c] = (lane < S) ? Ks[lane * HD + c] : 0.f;
        vreg[c] = (lane < S) ? Vs[lane * HD + c] : 0.f;
    }
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    for (int i = w; i < S; i += 4) {
        float s = 0.f, dpd = 0.f;
#pragma unroll
        for (int c4 = 0; c4 < HD4; ++c4) {
            const float4 q = *reinterpret_cast<const float4*>(&Qs[i * HD + c4 * 4]);
            const float4 g = *reinterpret_cast<const float4*>(&Os[i * HD + c4 * 4]);
            s += q.x * kreg[c4 * 4 + 0] + q.y * kreg[c4 * 4 + 1] + q.z * kreg[c4 * 4 + 2] + q.w * kreg[c4 * 4 + 3];
            dpd += g.x * vreg[c4 * 4 + 0] + g.y * vreg[c4 * 4 + 1] + g.z * vreg[c4 * 4 + 2] + g.w * vreg[c4 * 4 + 3];
        }
        const float l = lse[((long)b * H + h) * S + i];
        const float p = (lane < S) ? expf(s * scale - l) : 0.f;
        float mask = 1.f;
        if (drop.p > 0.f && lane < S)
            mask = cst_drop_mask(drop, dseed, (uint32_t)((((long)b * H + h) * S + i) * S + lane));
        const float dp = (lane < S) ? dpd * mask : 0.f;
        const float delta = wave_sum(dp * p);
        if (lane < S) {
            const float dsv = p * (dp - delta) * scale;
            Dm[i * SP + lane] = dsv;
            DmT[lane * SP + i] = dsv;
            PmT[lane * SP + i] = p * mask;
        }
    }
    __syncthreads();
    // pass 2: lane = head-dim column c; each wave keeps its rows i = w, w+4, ... as accumulators and
    // walks the keys 4 at a time: one broadcast b128 of the S x S image per row + 4 operand reads
    // per 4 keys feed 4*rows FMAs.
    float* dq = dqkv + (long)b * S * 3 * d + h * HD;
    const int S4 = (S + 3) / 4;
    for (int c = lane; c < HD; c += 64) {
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            const float* img = which == 0 ? Dm : (which == 1 ? DmT : PmT);     // dQ = dS K ; dK = dS^T Q ; dV = Pd^T dO
            const float* opr = which == 0 ? Ks : (which == 1 ? Qs : Os);
            float acc[RMAX];
#pragma unroll
            for (int r = 0; r < RMAX; ++r) acc[r] = 0.f;
            for (int j4 = 0; j4 < S4; ++j4) {
                const int j = j4 * 4;
                const float o0 = opr[j * HD + c];
                const float o1 = (j + 1 < S) ? opr[(j + 1) * HD + c] : 0.f;
                const float o2 = (j + 2 < S) ? opr[(j + 2) * HD + c] : 0.f;
                const float o3 = (j + 3 < S) ? opr[(j + 3) * HD + c] : 0.f;
#pragma unroll
                for (int r = 0; r < RMAX; ++r) {
                    const int i = w + 4 * r;
                    if (i < S) {
                        const float4 m4 = *reinterpret_cast<const float4*>(&img[i * SP + j]);
                        acc[r] += m4.x * o0 + m4.y * o1 + m4.z * o2 + m4.w * o3;
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < RMAX; ++r) {
                const int i = w + 4 * r;
                if (i < S) dq[(long)i * 3 * d + which * d + c] = acc[r];
            }
        }
    }
}

extern "C" int cst_mha_bwd(const float* qkv, const float* dout, const float* lse, float* dqkv,
                           int B, int S, int H, int hd,
                           float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                           void* stream) {
    CST_REQUIRE(qkv && dout && lse && dqkv, "cst_mha_bwd: null pointer");
    CST_REQUIRE(B > 0 && S > 0 && S <= MHA_SMAX && H > 0, "cst_mha_bwd: S=%d unsupported (max %d)", S, MHA_SMAX);
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev);
    const float scale = 1.0f / sqrtf((float)hd);
    const size_t lds = sizeof(float) * ((size_t)4 * S * hd + (size_t)3 * S * ((S + 3) / 4 * 4 + 4));
    CST_REQUIRE(lds <= 160 * 1024, "cst_mha_bwd: LDS need %zu exceeds 160 KiB", lds);
    dim3 grid(B * H), block(256);
    hipStream_t st = (hipStream_t)stream;
#define MHA_BWD_CASE(HDV)                                                                                         \
    case HDV: {                                                                                                   \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mha_bwd_kernel<HDV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((mha_bwd_kernel<HDV>), grid, block, lds, st, qkv, dout, lse, dqkv, S, H, scale, dr);   \
        break;                                                                                                    \
    }
    switch (hd) {
        MHA_BWD_CASE(8) MHA_BWD_CASE(16) MHA_BWD_CASE(32) MHA_BWD_CASE(64) MHA_BWD_CASE(96)
        default: cst_set_error("cst_mha_bwd: head dim %d unsupported (8, 16, 32, 64, 96)", hd); return CST_ERR_ARG;
    }
#undef MHA_BWD_CASE
    CST_LAUNCH_CHECK("cst_mha_bwd");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// Single-query attention, forward.  q [B, ldq] (D used), mem [B,L,D], out [B, ldo], p [B,L].
// The batch row's memory tile (L x D fp32, <= 64 x 512 -> 128 KiB) is pulled into LDS with every
// load in flight at once (one HBM/L2 round trip), then scores, softmax and the weighted sum run
// from LDS.  Optionally also writes the dropped copy [q | out] * mask of the decoder's i_ffn
// (rnn.py:78-79) so the step needs no separate dropout launch.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void dot_attn_fwd_kernel(const float* __restrict__ q, long ldq,
                                                           const float* __restrict__ mem, float* __restrict__ out, long ldo,
                                                           float* __restrict__ p, int L, int D, float scale,
                                                           float* __restrict__ dropped, long lddrop,
                                                           unsigned short* __restrict__ dropped_b, long lddropb, CstDrop drop) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* ms = dsm;                 // [L][D]
    float* qs = ms + L * D;          // [D]
    float* sc = qs + D;              // [64]
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* qb = q + (long)b * ldq;
    const float4* mb4 = reinterpret_cast<const float4*>(mem + (long)b * L * D);
    const int n4 = L * D / 4;
    for (int e = threadIdx.x; e < n4; e += blockDim.x) reinterpret_cast<float4*>(ms)[e] = mb4[e];
    for (int c = threadIdx.x; c < D; c += blockDim.x) qs[c] = qb[c];
    __syncthreads();
    for (int j = w; j < L; j += (int)(blockDim.x >> 6)) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += qs[c] * ms[j * D + c];
        s = wave_sum(s);
        if (lane == 0) sc[j] = s * scale;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = 0; j < L; ++j) m = fmaxf(m, sc[j]);
    float sum = 0.f;
    for (int j = 0; j < L; ++j) sum += expf(sc[j] - m);
    __syncthreads();
    if (threadIdx.x < L) {
        const float pj = expf(sc[threadIdx.x] - m) / sum;
        sc[threadIdx.x] = pj;
        p[(long)b * L + threadIdx.x] = pj;
    }
    __syncthreads();
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float o = 0.f;
        for (int j = 0; j < L; ++j) o += sc[j] * ms[j * D + c];
        out[(long)b * ldo + c] = o;
        if (dropped || dropped_b) {
            // dropout index space is the (B, 2D) matrix [q | out]
            const float mq = drop.p > 0.f ? cst_drop_mask(drop, dseed, (uint32_t)((long)b * 2 * D + c)) : 1.f;
            const float mo = drop.p > 0.f ? cst_drop_mask(drop, dseed, (uint32_t)((long)b * 2 * D + D + c)) : 1.f;
            if (dropped) {
                float* dr = dropped + (long)b * lddrop;
                dr[c] = qs[c] * mq;
                dr[D + c] = o * mo;
            }
            if (dropped_b) {                                       // bf16 copy: A operand of fn_1
                unsigned short* db = dropped_b + (long)b * lddropb;
                __bf16 h0 = (__bf16)(qs[c] * mq), h1 = (__bf16)(o * mo);
                db[c] = __builtin_bit_cast(unsigned short, h0);
                db[D + c] = __builtin_bit_cast(unsigned short, h1);
            }
        }
    }
}

extern "C" int cst_dot_attn_fwd(const float* q, long ldq, const float* mem, float* out, long ldo, float* p,
                                int B, int L, int D, float* dropped, long lddrop, void* dropped_bf16, long lddropb,
                                float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                                void* stream) {
    CST_REQUIRE(q && mem && out && p, "cst_dot_attn_fwd: null pointer");
    CST_REQUIRE(B > 0 && L > 0 && L <= MHA_SMAX && D > 0 && D % 4 == 0, "cst_dot_attn_fwd: L=%d (max %d) or D=%d unsupported", L, MHA_SMAX, D);
    const size_t lds = sizeof(float) * ((size_t)L * D + D + 64);
    CST_REQUIRE(lds <= 160 * 1024, "cst_dot_attn_fwd: memory tile of %zu bytes exceeds the 160 KiB LDS", lds);
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)dot_attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dot_attn_fwd_kernel, dim3(B), dim3(B <= 1024 ? 1024 : 256), lds, (hipStream_t)stream, q, ldq, mem, out, ldo, p, L, D,
                       1.0f / sqrtf((float)D), dropped, lddrop, (unsigned short*)dropped_bf16, lddropb, dr);
    CST_LAUNCH_CHECK("cst_dot_attn_fwd");
    return CST_OK;
}

// backward: dq (+)= sum_j ds_j mem_j ; dmem[j] += p_j dout + ds_j q   (dmem accumulates over steps)
__global__ __launch_bounds__(1024) void dot_attn_bwd_kernel(const float* __restrict__ dout, long lddo,
                                                           const float* __restrict__ q, long ldq,
                                                           const float* __restrict__ mem, const float* __restrict__ p,
                                                           float* __restrict__ dq, long lddq, int dq_accumulate,
                                                           float* __restrict__ dmem, int L, int D, float scale) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* ms = dsm;                 // [L][D]
    float* gs = ms + L * D;          // [D] dout row
    float* ds = gs + D;              // [64]
    float* ps = ds + 64;             // [64]
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const float* gb = dout + (long)b * lddo;
    const float* qb = q + (long)b * ldq;
    const float4* mb4 = reinterpret_cast<const float4*>(mem + (long)b * L * D);
    const int n4 = L * D / 4;
    for (int e = threadIdx.x; e < n4; e += blockDim.x) reinterpret_cast<float4*>(ms)[e] = mb4[e];
    for (int c = threadIdx.x; c < D; c += blockDim.x) gs[c] = gb[c];
    if (threadIdx.x < L) ps[threadIdx.x] = p[(long)b * L + threadIdx.x];
    __syncthreads();
    for (int j = w; j < L; j += (int)(blockDim.x >> 6)) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += gs[c] * ms[j * D + c];
        s = wave_sum(s);
        if (lane == 0) ds[j] = s;          // dp_j
    }
    __syncthreads();
    float delta = 0.f;
    for (int j = 0; j < L; ++j) delta += ds[j] * ps[j];
    __syncthreads();
    if (threadIdx.x < L) ds[threadIdx.x] = ps[threadIdx.x] * (ds[threadIdx.x] - delta) * scale;
    __syncthreads();
    float* dmb = dmem + (long)b * L * D;
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        const float g = gs[c], qc = qb[c];
        float a = 0.f;
        for (int j = 0; j < L; ++j) {
            a += ds[j] * ms[j * D + c];
            dmb[(long)j * D + c] += ps[j] * g + ds[j] * qc;
        }
        float* o = dq + (long)b * lddq + c;
        *o = dq_accumulate ? *o + a : a;
    }
}

extern "C" int cst_dot_attn_bwd(const float* dout, long lddo, const float* q, long ldq, const float* mem, const float* p,
                                float* dq, long lddq, int dq_accumulate, float* dmem, int B, int L, int D, void* stream) {
    CST_REQUIRE(dout && q && mem && p && dq && dmem, "cst_dot_attn_bwd: null pointer");
    CST_REQUIRE(B > 0 && L > 0 && L <= MHA_SMAX && D > 0 && D % 4 == 0, "cst_dot_attn_bwd: L=%d (max %d) or D=%d unsupported", L, MHA_SMAX, D);
    const size_t lds = sizeof(float) * ((size_t)L * D + D + 128);
    CST_REQUIRE(lds <= 160 * 1024, "cst_dot_attn_bwd: memory tile of %zu bytes exceeds the 160 KiB LDS", lds);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)dot_attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(dot_attn_bwd_kernel, dim3(B), dim3(B <= 1024 ? 1024 : 256), lds, (hipStream_t)stream, dout, lddo, q, ldq, mem, p,
                       dq, lddq, dq_accumulate, dmem, L, D, 1.0f / sqrtf((float)D));
    CST_LAUNCH_CHECK("cst_dot_attn_bwd");
    return CST_OK;
}
