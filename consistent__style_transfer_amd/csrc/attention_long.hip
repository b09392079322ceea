// Unmasked multi-head self-attention for 64 < S <= 128 keys (exact fp32), the "long" twins of
// mha_fwd_kernel / mha_bwd_kernel in attention.hip.
//
// Why they exist: the Matcher attends over cat(x1, x2) (match.py:36-39), so S = L1 + L2.  Yelp
// (max_len 18, arguments.py:40) stays under 64 even after transfer_noise lengthens sentences
// (data_util.py:45-53), the book corpus (max_len 30, arguments.py:43) does not: 60 before noise,
// ~80 after.  The S <= 64 kernels keep the whole S x S plane of a (batch row, head) in LDS, which
// stops fitting at S = 128 (274 KB for the backward at hd = 64).
//
// Same arithmetic, same exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32), same dropout index space
// (((b H + h) S + i) S + j); what changes is the blocking: K and V of the (b, h) pair stay resident
// in LDS, the queries go through in blocks of QP = 16 QT rows, and per block only a [QP][S] slab of
// the plane exists.  Forward: scores -> softmax -> P V per block.  Backward: per block dQ is
// complete (contraction over all keys), dK / dV are contractions over the QUERY index and are
// carried in registers across the blocks (each wave owns a fixed set of 16x16 output tiles).
#include "cst_common.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int MHL_NW = 8;
constexpr int MHL_NTHR = MHL_NW * 64;

__host__ __device__ inline size_t mhl_fwd_lds_floats(int ST, int hd, int QT) {
    const size_t SP = (size_t)ST * 16, QP = (size_t)QT * 16;
    return 2 * SP * (hd + 4) + QP * (hd + 4) + 16 + QP * (SP + 4);
}
__host__ __device__ inline size_t mhl_bwd_lds_floats(int ST, int hd, int QT) {
    const size_t SP = (size_t)ST * 16, QP = (size_t)QT * 16;
    return 2 * SP * (hd + 4) + 2 * QP * (hd + 4) + 16 + 2 * QP * (SP + 4) + QP * 8 + QP;
}

// rows [r0, r0 + ROWS) of a [S][3d]-strided head slice -> a padded [ROWS][HD + 4] LDS image (rows past S - 1
// replicate row S - 1); every 16-byte load is issued before the first LDS store
template <int HD, int ROWS, int NMAT>
__device__ __forceinline__ void mhl_stage(const float* const (&src)[NMAT], const long (&ld)[NMAT], float* const (&dst)[NMAT], int r0, int S) {
    constexpr int HD4 = HD / 4, HDS = HD + 4, nel = ROWS * HD4, IT = (nel + MHL_NTHR - 1) / MHL_NTHR;
    float4 t[NMAT][IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int e = min((int)threadIdx.x + MHL_NTHR * it, nel - 1);
        const int i = min(r0 + e / HD4, S - 1), c = (e % HD4) * 4;
#pragma unroll
        for (int m = 0; m < NMAT; ++m) t[m][it] = *reinterpret_cast<const float4*>(src[m] + (long)i * ld[m] + c);
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int e = threadIdx.x + MHL_NTHR * it;
        if (e < nel) {
            const int o = (e / HD4) * HDS + (e % HD4) * 4;
#pragma unroll
            for (int m = 0; m < NMAT; ++m) *reinterpret_cast<float4*>(dst[m] + o) = t[m][it];
        }
    }
}

// acc += A B^T over the head dimension for one 16x16 tile: a, b point at this lane's row and k segment
template <int HD>
__device__ __forceinline__ f32x4_t mhl_dot_hd(const float* a, const float* b) {
    constexpr int SEG = HD / 4;
    f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (SEG % 4 == 0) {
#pragma unroll
        for (int k4 = 0; k4 < SEG / 4; ++k4) {
            const float4 x = *reinterpret_cast<const float4*>(a + k4 * 4);
            const float4 y = *reinterpret_cast<const float4*>(b + k4 * 4);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.x, y.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.y, y.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.z, y.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x.w, y.w, acc1, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int k = 0; k < SEG; ++k) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[k], b[k], acc0, 0, 0, 0);
    }
    return acc0 + acc1;
}

// ---------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------
template <int HD, int ST, int QT>
__global__ __launch_bounds__(MHL_NTHR) void mha_fwd_long_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                                float* __restrict__ lse, int S, int H, float scale, CstDrop drop,
                                                                unsigned short* __restrict__ outb, long ldob) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HDS = HD + 4, SEG = HD / 4, NT = (HD + 15) / 16;
    constexpr int SP = ST * 16, SS = SP + 4, KSEG = SP / 4, QP = QT * 16;
    float* Ks = smem;                    // [SP][HDS]
    float* Vs = Ks + SP * HDS;
    float* Qs = Vs + SP * HDS;           // [QP][HDS]
    float* Pm = Qs + QP * HDS + 16;      // [QP][SS]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int d = H * HD;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const float* base = qkv + (long)b * S * 3 * d + h * HD;
    {
        const float* const src[2] = {base + d, base + 2 * d};
        const long ld[2] = {3L * d, 3L * d};
        float* const dst[2] = {Ks, Vs};
        mhl_stage<HD, SP, 2>(src, ld, dst, 0, S);
    }
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    float* ob = out + (long)b * S * d + h * HD;
    for (int q0 = 0; q0 < S; q0 += QP) {
        {
            const float* const src[1] = {base};
            const long ld[1] = {3L * d};
            float* const dst[1] = {Qs};
            mhl_stage<HD, QP, 1>(src, ld, dst, q0, S);
        }
        __syncthreads();
        // ---- phase A: scores of this query block against every key
        for (int t = w; t < QT * ST; t += MHL_NW) {
            const int mt = t / ST, nt = t % ST;
            const f32x4_t acc = mhl_dot_hd<HD>(Qs + (mt * 16 + lr) * HDS + lq * SEG, Ks + (nt * 16 + lr) * HDS + lq * SEG);
            const int j = nt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) Pm[(mt * 16 + lq * 4 + r) * SS + j] = acc[r] * scale;
        }
        __syncthreads();
        // ---- phase B: row softmax, one row per 16-lane group
        for (int il0 = 4 * w; il0 < QP; il0 += 4 * MHL_NW) {
            const int il = il0 + lq, i = q0 + il;
            const bool row_ok = i < S;
            float sv[ST];
            float m = -INFINITY;
#pragma unroll
            for (int n = 0; n < ST; ++n) {
                const int j = lr + 16 * n;
                sv[n] = j < S ? Pm[il * SS + j] : -INFINITY;
                m = fmaxf(m, sv[n]);
            }
            m = row16_max(m);
            float sum = 0.f;
#pragma unroll
            for (int n = 0; n < ST; ++n) {
                sv[n] = (lr + 16 * n < S) ? expf(sv[n] - m) : 0.f;
                sum += sv[n];
            }
            sum = row16_sum(sum);
            const float inv = 1.f / sum;
            if (row_ok && lr == 0) lse[((long)b * H + h) * S + i] = m + logf(sum);
#pragma unroll
            for (int n = 0; n < ST; ++n) {
                const int j = lr + 16 * n;
                float pv = row_ok ? sv[n] * inv : 0.f;
                if (drop.p > 0.f && row_ok && j < S)
                    pv *= cst_drop_mask(drop, dseed, (uint32_t)((((long)b * H + h) * S + i) * S + j));
                Pm[il * SS + j] = pv;                     // keys >= S and rows >= S: exactly 0
            }
        }
        __syncthreads();
        // ---- phase C: O[i][c] = sum_j Pd[i][j] V[j][c]
        constexpr int PAIR = (NT % 2 == 0) ? 2 : 1, NP = NT / PAIR;
        for (int u = w; u < QT * NP; u += MHL_NW) {
            const int mt = u / NP, n0 = (u - mt * NP) * PAIR * 16;
            const float* a = Pm + (mt * 16 + lr) * SS + lq * KSEG;
            const float* bp = Vs + (lq * KSEG) * HDS + n0 + lr;
            f32x4_t acc[PAIR];
#pragma unroll
            for (int q = 0; q < PAIR; ++q) acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int k = 0; k < KSEG; ++k) {
                const float av = a[k];
#pragma unroll
                for (int q = 0; q < PAIR; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[k * HDS + q * 16], acc[q], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < PAIR; ++q) {
                const int n = n0 + q * 16 + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mrow = q0 + mt * 16 + lq * 4 + r;
                    if (mrow < S && n < HD) {
                        ob[(long)mrow * d + n] = acc[q][r];
                        if (outb) {
                            __bf16 hh = (__bf16)acc[q][r];
                            outb[((long)b * S + mrow) * ldob + h * HD + n] = __builtin_bit_cast(unsigned short, hh);
                        }
                    }
                }
            }
        }
        __syncthreads();                                  // Qs / Pm are rewritten by the next block
    }
}

// ---------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------
template <int HD, int ST, int QT>
__global__ __launch_bounds__(MHL_NTHR) void mha_bwd_long_kernel(const float* __restrict__ qkv, const float* __restrict__ dout,
                                                                const float* __restrict__ lse, float* __restrict__ dqkv,
                                                                int S, int H, float scale, CstDrop drop,
                                                                unsigned short* __restrict__ dqkvb, long lddb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HDS = HD + 4, SEG = HD / 4, NT = (HD + 15) / 16;
    constexpr int SP = ST * 16, SS = SP + 4, KSEG = SP / 4, QP = QT * 16, KQ = QP / 4;
    float* Ks = smem;                    // [SP][HDS]
    float* Vs = Ks + SP * HDS;
    float* Qs = Vs + SP * HDS;           // [QP][HDS]
    float* Os = Qs + QP * HDS;           // [QP][HDS]  dO block
    float* Pm = Os + QP * HDS + 16;      // [QP][SS]   Pd
    float* Dm = Pm + QP * SS;            // [QP][SS]   dS
    float* part = Dm + QP * SS;          // [QP][8]    row sums of dP o P per key tile
    float* lse_s = part + QP * 8;        // [QP]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int d = H * HD;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const float* base = qkv + (long)b * S * 3 * d + h * HD;
    const float* dob = dout + (long)b * S * d + h * HD;
    float* dq = dqkv + (long)b * S * 3 * d + h * HD;
    {
        const float* const src[2] = {base + d, base + 2 * d};
        const long ld[2] = {3L * d, 3L * d};
        float* const dst[2] = {Ks, Vs};
        mhl_stage<HD, SP, 2>(src, ld, dst, 0, S);
    }
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    constexpr int PAIR = (NT % 2 == 0) ? 2 : 1, NP = NT / PAIR;
    constexpr int NKV = 2 * ST * NP, IPW = (NKV + MHL_NW - 1) / MHL_NW;     // dK and dV work items per wave
    f32x4_t kv[IPW][PAIR];
#pragma unroll
    for (int n = 0; n < IPW; ++n)
#pragma unroll
        for (int q = 0; q < PAIR; ++q) kv[n][q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int q0 = 0; q0 < S; q0 += QP) {
        {
            const float* const src[2] = {base, dob};
            const long ld[2] = {3L * d, (long)d};
            float* const dst[2] = {Qs, Os};
            mhl_stage<HD, QP, 2>(src, ld, dst, q0, S);
        }
        if (threadIdx.x < QP) lse_s[threadIdx.x] = (q0 + (int)threadIdx.x) < S ? lse[((long)b * H + h) * S + q0 + threadIdx.x] : 0.f;
        __syncthreads();
        // ---- phase A -------------------------------------------------------------------------
        constexpr int ntile = QT * ST, TPW = (ntile + MHL_NW - 1) / MHL_NW;
        f32x4_t Pf[TPW], Df[TPW];
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int t = w + MHL_NW * tt;
            Pf[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            Df[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            if (t < ntile) {
                const int mt = t / ST, nt = t % ST;
                const f32x4_t accS = mhl_dot_hd<HD>(Qs + (mt * 16 + lr) * HDS + lq * SEG, Ks + (nt * 16 + lr) * HDS + lq * SEG);
                const f32x4_t accD = mhl_dot_hd<HD>(Os + (mt * 16 + lr) * HDS + lq * SEG, Vs + (nt * 16 + lr) * HDS + lq * SEG);
                const int j = nt * 16 + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int il = mt * 16 + lq * 4 + r, i = q0 + il;
                    const bool valid = i < S && j < S;
                    const float pv = valid ? expf(accS[r] * scale - lse_s[il]) : 0.f;
                    float mask = 1.f;
                    if (drop.p > 0.f && valid)
                        mask = cst_drop_mask(drop, dseed, (uint32_t)((((long)b * H + h) * S + i) * S + j));
                    const float dp = valid ? accD[r] * mask : 0.f;
                    Pm[il * SS + j] = pv * mask;
                    float rs = dp * pv;
                    rs = row16_sum(rs);
                    if (lr == 0) part[il * 8 + nt] = rs;
                    Pf[tt][r] = pv;
                    Df[tt][r] = dp;
                }
            }
        }
        __syncthreads();
        // ---- phase B -------------------------------------------------------------------------
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt) {
            const int t = w + MHL_NW * tt;
            if (t < ntile) {
                const int mt = t / ST, nt = t % ST;
                const int j = nt * 16 + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int il = mt * 16 + lq * 4 + r;
                    float delta = 0.f;
#pragma unroll
                    for (int n = 0; n < ST; ++n) delta += part[il * 8 + n];
                    Dm[il * SS + j] = Pf[tt][r] * (Df[tt][r] - delta) * scale;
                }
            }
        }
        __syncthreads();
        // ---- phase C1: dQ[i][c] = sum_j dS[i][j] K[j][c], complete for this block --------------
        for (int u = w; u < QT * NP; u += MHL_NW) {
            const int mt = u / NP, n0 = (u - mt * NP) * PAIR * 16;
            const float* a = Dm + (mt * 16 + lr) * SS + lq * KSEG;
            const float* bp = Ks + (lq * KSEG) * HDS + n0 + lr;
            f32x4_t acc[PAIR];
#pragma unroll
            for (int q = 0; q < PAIR; ++q) acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int k = 0; k < KSEG; ++k) {
                const float av = a[k];
#pragma unroll
                for (int q = 0; q < PAIR; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[k * HDS + q * 16], acc[q], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < PAIR; ++q) {
                const int n = n0 + q * 16 + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = q0 + mt * 16 + lq * 4 + r;
                    if (m < S && n < HD) {
                        dq[(long)m * 3 * d + n] = acc[q][r];
                        if (dqkvb) {
                            __bf16 hh = (__bf16)acc[q][r];
                            dqkvb[((long)b * S + m) * lddb + h * HD + n] = __builtin_bit_cast(unsigned short, hh);
                        }
                    }
                }
            }
        }
        // ---- phase C2: dK[j][c] += sum_{i in block} dS[i][j] Q[i][c], dV[j][c] += sum Pd[i][j] dO[i][c] ----
#pragma unroll
        for (int n = 0; n < IPW; ++n) {
            const int u = w + MHL_NW * n;
            if (u < NKV) {
                const int which = u / (ST * NP), rem = u - which * (ST * NP);
                const int jt = rem / NP, n0 = (rem - jt * NP) * PAIR * 16;
                const float* a = (which == 0 ? Dm : Pm) + (lq * KQ) * SS + jt * 16 + lr;      // A[m = j][k = i]
                const float* bp = (which == 0 ? Qs : Os) + (lq * KQ) * HDS + n0 + lr;         // B[k = i][n = c]
#pragma unroll
                for (int k = 0; k < KQ; ++k) {
                    const float av = a[k * SS];
#pragma unroll
                    for (int q = 0; q < PAIR; ++q)
                        kv[n][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[k * HDS + q * 16], kv[n][q], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                  // Qs / Os / Pm / Dm are rewritten by the next block
    }
#pragma unroll
    for (int n = 0; n < IPW; ++n) {
        const int u = w + MHL_NW * n;
        if (u < NKV) {
            const int which = u / (ST * NP), rem = u - which * (ST * NP);
            const int jt = rem / NP, n0 = (rem - jt * NP) * PAIR * 16;
#pragma unroll
            for (int q = 0; q < PAIR; ++q) {
                const int c = n0 + q * 16 + lr;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = jt * 16 + lq * 4 + r;
                    if (m < S && c < HD) {
                        dq[(long)m * 3 * d + (1 + which) * d + c] = kv[n][q][r];
                        if (dqkvb) {
                            __bf16 hh = (__bf16)kv[n][q][r];
                            dqkvb[((long)b * S + m) * lddb + (1 + which) * d + h * HD + c] = __builtin_bit_cast(unsigned short, hh);
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// launchers (called from cst_mha_fwd_b / cst_mha_bwd_b in attention.hip when S > 64)
// ---------------------------------------------------------------------------------------------
static const size_t MHL_LDS_MAX = 160 * 1024;

template <int HD, int ST, int QT>
static int mhl_fwd_go(const float* qkv, float* out, float* lse, int B, int S, int H, float scale, CstDrop dr, void* outb, long ldob, hipStream_t st) {
    const size_t lds = sizeof(float) * mhl_fwd_lds_floats(ST, HD, QT);
    if (lds > MHL_LDS_MAX) { cst_set_error("cst_mha_fwd: S=%d hd=%d needs %zu bytes of LDS", S, HD, lds); return CST_ERR_ARG; }
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mha_fwd_long_kernel<HD, ST, QT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mha_fwd_long_kernel<HD, ST, QT>), dim3(B * H), dim3(MHL_NTHR), lds, st, qkv, out, lse, S, H, scale, dr, (unsigned short*)outb, ldob);
    return CST_OK;
}
template <int HD, int ST, int QT>
static int mhl_bwd_go(const float* qkv, const float* dout, const float* lse, float* dqkv, int B, int S, int H, float scale, CstDrop dr,
                      void* dqb, long lddb, hipStream_t st) {
    const size_t lds = sizeof(float) * mhl_bwd_lds_floats(ST, HD, QT);
    if (lds > MHL_LDS_MAX) { cst_set_error("cst_mha_bwd: S=%d hd=%d needs %zu bytes of LDS", S, HD, lds); return CST_ERR_ARG; }
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mha_bwd_long_kernel<HD, ST, QT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((mha_bwd_long_kernel<HD, ST, QT>), dim3(B * H), dim3(MHL_NTHR), lds, st, qkv, dout, lse, dqkv, S, H, scale, dr,
                       (unsigned short*)dqb, lddb);
    return CST_OK;
}

int cst_mha_long_max_s() { return 128; }

// S in (64, 96] runs the 6-tile instantiation, (96, 128] the 8-tile one.  Head dims: the reference's 64 (d 512 / 8 heads), 96
// (BASELINE configs[3]: d 768 / 8 heads) and the 8 of the tiny parity configuration.
int cst_mha_fwd_long(const float* qkv, float* out, float* lse, int B, int S, int H, int hd, float scale, CstDrop dr,
                     void* outb, long ldob, hipStream_t st) {
    const bool six = S <= 96;
    switch (hd) {
        case 8: return six ? mhl_fwd_go<8, 6, 4>(qkv, out, lse, B, S, H, scale, dr, outb, ldob, st) : mhl_fwd_go<8, 8, 4>(qkv, out, lse, B, S, H, scale, dr, outb, ldob, st);
        case 64: return six ? mhl_fwd_go<64, 6, 4>(qkv, out, lse, B, S, H, scale, dr, outb, ldob, st) : mhl_fwd_go<64, 8, 4>(qkv, out, lse, B, S, H, scale, dr, outb, ldob, st);
        case 96: return six ? mhl_fwd_go<96, 6, 2>(qkv, out, lse, B, S, H, scale, dr, outb, ldob, st) : mhl_fwd_go<96, 8, 2>(qkv, out, lse, B, S, H, scale, dr, outb, ldob, st);
        default: cst_set_error("cst_mha_fwd: head dim %d unsupported for S > 64 (8, 64, 96)", hd); return CST_ERR_ARG;
    }
}
int cst_mha_bwd_long(const float* qkv, const float* dout, const float* lse, float* dqkv, int B, int S, int H, int hd, float scale, CstDrop dr,
                     void* dqb, long lddb, hipStream_t st) {
    const bool six = S <= 96;
    switch (hd) {
        case 8: return six ? mhl_bwd_go<8, 6, 2>(qkv, dout, lse, dqkv, B, S, H, scale, dr, dqb, lddb, st) : mhl_bwd_go<8, 8, 2>(qkv, dout, lse, dqkv, B, S, H, scale, dr, dqb, lddb, st);
        case 64: return six ? mhl_bwd_go<64, 6, 2>(qkv, dout, lse, dqkv, B, S, H, scale, dr, dqb, lddb, st) : mhl_bwd_go<64, 8, 2>(qkv, dout, lse, dqkv, B, S, H, scale, dr, dqb, lddb, st);
        case 96: return six ? mhl_bwd_go<96, 6, 2>(qkv, dout, lse, dqkv, B, S, H, scale, dr, dqb, lddb, st) : mhl_bwd_go<96, 8, 2>(qkv, dout, lse, dqkv, B, S, H, scale, dr, dqb, lddb, st);
        default: cst_set_error("cst_mha_bwd: head dim %d unsupported for S > 64 (8, 64, 96)", hd); return CST_ERR_ARG;
    }
}
