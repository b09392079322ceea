// Attention kernels for gfx950 (exact fp32; tiles are tiny: S <= 64 keys).
//
//  (1) Unmasked multi-head self-attention core of nn.TransformerEncoderLayer as the reference
//      configures it (mlm.py:20-22, match.py:18-20; no mask is ever passed, mlm.py:43,
//      match.py:39): per (batch row, head), P = softmax(Q K^T / sqrt(hd)), attention dropout on
//      P, O = P V.  S = L (MLM) or L1+L2 (Matcher) <= 64.
//      One 8-wave workgroup per (b, h); every product of the forward (Q K^T, P V) and of the
//      backward (Q K^T, dO V^T, Pd^T dO, dS K, dS^T Q) runs on the exact-fp32 matrix pipe
//      (v_mfma_f32_16x16x4_f32) from padded LDS images of Q, K, V, dO and of the S x S planes;
//      softmax statistics are 16-lane DPP reductions.
//  (2) Single-query dot-product attention of the generator's decoder (rnn.py:46-50, called at
//      rnn.py:76): one workgroup per batch row, memory (L' <= 64 rows of D) streamed once.
#include "cst_common.h"

#define MHA_SMAX 64                  // single-tile kernels below; 64 < S <= 128 goes to attention_long.hip
#define MHA_SMAX_LONG 128

int cst_mha_fwd_long(const float* qkv, float* out, float* lse, int B, int S, int H, int hd, float scale, CstDrop dr,
                     void* outb, long ldob, hipStream_t st);
int cst_mha_bwd_long(const float* qkv, const float* dout, const float* lse, float* dqkv, int B, int S, int H, int hd, float scale, CstDrop dr,
                     void* dqb, long lddb, hipStream_t st);

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int MHA_NW = 8;            // waves per workgroup: 2-3 resident workgroups give 4-6 waves / SIMD

// four consecutive elements at element offset `off` of an fp32 (HB = false) or bf16 (HB = true) array, as floats.  The bf16 forms
// (cst_mha_fwd_h / cst_mha_bwd_h) halve the HBM streams of the attention core in bf16 mode -- qkv comes straight from the
// in-projection GEMM's bf16 output, d(attention output) from the out-projection dgrad's -- while the LDS images and every
// product stay fp32 (exact arithmetic on bf16-rounded inputs).
template <bool HB>
__device__ __forceinline__ float4 mha_ld4(const void* base, long off) {
    if constexpr (HB) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    } else {
        return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
    }
}

// ---------------------------------------------------------------------------------------------
// MHA forward.  qkv [B,S,3d] (q | k | v thirds, heads interleaved inside each third as torch's
// packed in_proj does), out [B,S,d], lse [B,H,S].
//
// Both products run on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32 == an fmaf chain).  One
// workgroup of 8 waves per (batch row, head), S padded to SP = 16*ST rows (pad rows replicate row
// S-1, pad keys get probability 0):
//   phase A  Sc = scale * Q K^T per 16x16 tile (k split in four contiguous segments, one per
//            16-lane group: operands are b128 row reads of the padded [SP][HD+4] images) -> LDS
//   phase B  row softmax, one row per 16-lane group (DPP reductions), lse, attention dropout; Pd -> LDS
//   phase C  O = Pd V per 16-row block and pair of 16-column tiles
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline size_t mha_fwd_lds_floats(int S, int hd) {
    const size_t SP = (size_t)((S + 15) / 16) * 16;
    return 3 * SP * (hd + 4) + 16 + SP * (SP + 4);
}

template <int HD, int ST, bool QB = false>
__global__ __launch_bounds__(MHA_NW * 64) void mha_fwd_kernel(const void* __restrict__ qkv, float* __restrict__ out,
                                                             float* __restrict__ lse, int S, int H, float scale, CstDrop drop,
                                                             unsigned short* __restrict__ outb, long ldob) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HD4 = HD / 4, HDS = HD + 4, SEG = HD / 4, NT = (HD + 15) / 16;
    constexpr int SP = ST * 16, SS = SP + 4, KSEG = SP / 4, NTHR = MHA_NW * 64;
    float* Qs = smem;                    // [SP][HDS]
    float* Ks = Qs + SP * HDS;
    float* Vs = Ks + SP * HDS;
    float* Pm = Vs + SP * HDS + 16;      // [SP][SS]  scores, then dropped probabilities
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int d = H * HD;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const long base = (long)b * S * 3 * d + h * HD;          // element offset of this (b, h) slice
    {
        constexpr int nel = SP * HD4;
        constexpr int QIT = (nel + NTHR - 1) / NTHR;
        float4 tq[QIT], tk[QIT], tv[QIT];
#pragma unroll
        for (int it = 0; it < QIT; ++it) {
            const int e = min((int)threadIdx.x + NTHR * it, nel - 1);
            const int i = min(e / HD4, S - 1), c = (e % HD4) * 4;
            const long src = base + (long)i * 3 * d + c;
            tq[it] = mha_ld4<QB>(qkv, src);
            tk[it] = mha_ld4<QB>(qkv, src + d);
            tv[it] = mha_ld4<QB>(qkv, src + 2 * d);
        }
#pragma unroll
        for (int it = 0; it < QIT; ++it) {
            const int e = threadIdx.x + NTHR * it;
            if (e < nel) {
                const int o = (e / HD4) * HDS + (e % HD4) * 4;
                *reinterpret_cast<float4*>(&Qs[o]) = tq[it];
                *reinterpret_cast<float4*>(&Ks[o]) = tk[it];
                *reinterpret_cast<float4*>(&Vs[o]) = tv[it];
            }
        }
    }
    __syncthreads();
    // ---- phase A -----------------------------------------------------------------------------
    constexpr int ntile = ST * ST;
    for (int t = w; t < ntile; t += MHA_NW) {
        const int mt = t / ST, nt = t % ST;
        const float* qa = Qs + (mt * 16 + lr) * HDS + lq * SEG;
        const float* ka = Ks + (nt * 16 + lr) * HDS + lq * SEG;
        f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};       // two chains over alternate k
        if constexpr (SEG % 4 == 0) {
#pragma unroll
            for (int k4 = 0; k4 < SEG / 4; ++k4) {
                const float4 q4 = *reinterpret_cast<const float4*>(qa + k4 * 4);
                const float4 kx = *reinterpret_cast<const float4*>(ka + k4 * 4);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.x, kx.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.y, kx.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.z, kx.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.w, kx.w, acc1, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int k = 0; k < SEG; ++k) acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[k], ka[k], acc0, 0, 0, 0);
        }
        const int j = nt * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) Pm[(mt * 16 + lq * 4 + r) * SS + j] = (acc0[r] + acc1[r]) * scale;
    }
    __syncthreads();
    // ---- phase B: 16-lane group g of wave w owns rows i = 4 * (w + NW * pass) + g; lane lr holds keys lr + 16 n
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    for (int i0 = 4 * w; i0 < S; i0 += 4 * MHA_NW) {
        const int i = i0 + lq;
        const bool row_ok = i < S;
        const int ic = row_ok ? i : S - 1;
        float sv[ST];
        float m = -INFINITY;
#pragma unroll
        for (int n = 0; n < ST; ++n) {
            const int j = lr + 16 * n;
            sv[n] = j < S ? Pm[ic * SS + j] : -INFINITY;
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
        if (row_ok) {
            if (lr == 0) lse[((long)b * H + h) * S + i] = m + logf(sum);
#pragma unroll
            for (int n = 0; n < ST; ++n) {
                const int j = lr + 16 * n;
                float pv = sv[n] * inv;
                if (drop.p > 0.f && j < S)
                    pv *= cst_drop_mask(drop, dseed, (uint32_t)((((long)b * H + h) * S + i) * S + j));
                Pm[i * SS + j] = pv;                  // keys >= S: exactly 0
            }
        }
    }
    __syncthreads();
    // ---- phase C: O[i][c] = sum_j Pd[i][j] V[j][c]
    float* ob = out + (long)b * S * d + h * HD;
    constexpr int PAIR = (NT % 2 == 0) ? 2 : 1, NP = NT / PAIR;
    constexpr int nout = ST * NP;
    for (int u = w; u < nout; u += MHA_NW) {
        const int mt = u / NP, n0 = (u - mt * NP) * PAIR * 16;
        const float* a = Pm + (mt * 16 + lr) * SS + lq * KSEG;
        const float* bp = Vs + (lq * KSEG) * HDS + n0 + lr;
        f32x4_t acc[PAIR];
#pragma unroll
        for (int q = 0; q < PAIR; ++q) acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
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
                const int mrow = mt * 16 + lq * 4 + r;
                if (mrow < S && n < HD) {
                    if (out) ob[(long)mrow * d + n] = acc[q][r];
                    if (outb) {                                   // A operand of the out-projection
                        __bf16 hh = (__bf16)acc[q][r];
                        outb[((long)b * S + mrow) * ldob + h * HD + n] = __builtin_bit_cast(unsigned short, hh);
                    }
                }
            }
        }
    }
}

extern "C" int cst_mha_fwd_b(const float* qkv, float* out, float* lse, int B, int S, int H, int hd,
                             float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                             void* out_bf16, long ldob, void* stream);
extern "C" int cst_mha_fwd(const float* qkv, float* out, float* lse, int B, int S, int H, int hd,
                           float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                           void* stream) {
    return cst_mha_fwd_b(qkv, out, lse, B, S, H, hd, drop_p, drop_seed, drop_stream, drop_seed_dev, nullptr, 0, stream);
}

// ---------------------------------------------------------------------------------------------
// MHA forward, bf16 qkv (cst_mha_fwd_h), HD a multiple of 32: bf16 LDS images of Q, K, V (40 KiB in all at S = 36, hd = 96 against
// 68 KiB: four workgroups per CU instead of two), Q K^T on v_mfma_f32_16x16x32_bf16 (exact products, another summation order than the
// fp32-image kernel), P V on the fp32 matrix pipe with V widened from its image.  Mirrors mha_bwd_hb_kernel below.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 mha_bf16x8_t;

__host__ __device__ inline size_t mha_fwd_hb_lds_bytes(int S, int hd) {
    const size_t SP = (size_t)((S + 15) / 16) * 16;
    return 3 * SP * (hd + 8) * 2 + 64 + (sizeof(float) + 2) * SP * (SP + 4);
}

typedef __attribute__((ext_vector_type(4))) short mha_s16x4_t;
// ds_read_b64_tr_b16: the 16 lanes of a group read a 4-row x 16-column block of bf16 (lane 4q + p supplies the address of row q, columns
// 4p .. 4p + 3) and each lane receives COLUMN (lane % 16) of it, rows 0 .. 3 -- the MFMA operand of a product whose contraction index
// is the image's row index, without a transposed copy.  EXEC must be all ones (cdna_hip_programming.md T10).
// two / three transposed reads and ONE wait in one statement (the compiler believes an asm's outputs are valid when the statement ends)
__device__ __forceinline__ void mha_tr_read3(const unsigned short* p0, const unsigned short* p1, const unsigned short* p2,
                                             mha_s16x4_t& v0, mha_s16x4_t& v1, mha_s16x4_t& v2) {
    const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)p0;
    const unsigned a1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)p1;
    const unsigned a2 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)p2;
    asm volatile("ds_read_b64_tr_b16 %0, %3\n\tds_read_b64_tr_b16 %1, %4\n\tds_read_b64_tr_b16 %2, %5\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1), "=&v"(v2) : "v"(a0), "v"(a1), "v"(a2) : "memory");
}
__device__ __forceinline__ void mha_tr_read2(const unsigned short* p0, const unsigned short* p1, mha_s16x4_t& v0, mha_s16x4_t& v1) {
    const unsigned a0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)p0;
    const unsigned a1 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)p1;
    asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v0), "=&v"(v1) : "v"(a0), "v"(a1) : "memory");
}
__device__ __forceinline__ mha_s16x4_t mha_tr_read(const unsigned short* p) {
    mha_s16x4_t v;
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const unsigned short*)p;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

template <int HD, int ST>
__global__ __launch_bounds__(MHA_NW * 64) void mha_fwd_hb_kernel(const unsigned short* __restrict__ qkv, float* __restrict__ out,
                                                         float* __restrict__ lse, int S, int H, float scale, CstDrop drop,
                                                         unsigned short* __restrict__ outb, long ldob) {
    static_assert(HD % 32 == 0, "bf16 MFMA k-steps of 32");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int HD4 = HD / 4, HDB = HD + 8, NT = HD / 16;
    constexpr int SP = ST * 16, SS = SP + 4, KSEG = SP / 4, NTHR = MHA_NW * 64;
    unsigned short* Qs = reinterpret_cast<unsigned short*>(smem_raw);      // [SP][HDB]
    unsigned short* Ks = Qs + SP * HDB;
    unsigned short* Vs = Ks + SP * HDB;
    float* Pm = reinterpret_cast<float*>(Vs + SP * HDB + 32);               // [SP][SS]  scores, then Pd
    unsigned short* Pb = reinterpret_cast<unsigned short*>(Pm + SP * SS);   // [SP][SS]  bf16 Pd: the A operand of Pd V
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int d = H * HD;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const long base = (long)b * S * 3 * d + h * HD;
    {
        constexpr int nel = SP * HD4;
        constexpr int QIT = (nel + NTHR - 1) / NTHR;
        uint2 tq[QIT], tk[QIT], tv[QIT];
#pragma unroll
        for (int it = 0; it < QIT; ++it) {
            const int e = min((int)threadIdx.x + NTHR * it, nel - 1);
            const int i = min(e / HD4, S - 1), c = (e % HD4) * 4;
            const long src = base + (long)i * 3 * d + c;
            tq[it] = *reinterpret_cast<const uint2*>(qkv + src);
            tk[it] = *reinterpret_cast<const uint2*>(qkv + src + d);
            tv[it] = *reinterpret_cast<const uint2*>(qkv + src + 2 * d);
        }
#pragma unroll
        for (int it = 0; it < QIT; ++it) {
            const int e = threadIdx.x + NTHR * it;
            if (e < nel) {
                const int o = (e / HD4) * HDB + (e % HD4) * 4;
                *reinterpret_cast<uint2*>(&Qs[o]) = tq[it];
                *reinterpret_cast<uint2*>(&Ks[o]) = tk[it];
                *reinterpret_cast<uint2*>(&Vs[o]) = tv[it];
            }
        }
    }
    __syncthreads();
    // ---- phase A: scores, HD / 32 bf16 MFMAs per 16 x 16 tile ----------------------------------
    constexpr int ntile = ST * ST;
    for (int t = w; t < ntile; t += MHA_NW) {
        const int mt = t / ST, nt = t % ST;
        const unsigned short* qa = Qs + (mt * 16 + lr) * HDB + lq * 8;
        const unsigned short* ka = Ks + (nt * 16 + lr) * HDB + lq * 8;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const mha_bf16x8_t*>(qa + ks * 32),
                                                          *reinterpret_cast<const mha_bf16x8_t*>(ka + ks * 32), acc, 0, 0, 0);
        const int j = nt * 16 + lr;
#pragma unroll
        for (int r = 0; r < 4; ++r) Pm[(mt * 16 + lq * 4 + r) * SS + j] = acc[r] * scale;
    }
    __syncthreads();
    // ---- phase B: row softmax, lse, attention dropout (as mha_fwd_kernel) -----------------------
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    for (int i0 = 4 * w; i0 < S; i0 += 4 * MHA_NW) {
        const int i = i0 + lq;
        const bool row_ok = i < S;
        const int ic = row_ok ? i : S - 1;
        float sv[ST];
        float m = -INFINITY;
#pragma unroll
        for (int n = 0; n < ST; ++n) {
            const int j = lr + 16 * n;
            sv[n] = j < S ? Pm[ic * SS + j] : -INFINITY;
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
        if (row_ok) {
            if (lr == 0) lse[((long)b * H + h) * S + i] = m + logf(sum);
#pragma unroll
            for (int n = 0; n < ST; ++n) {
                const int j = lr + 16 * n;
                float pv = sv[n] * inv;
                if (drop.p > 0.f && j < S)
                    pv *= cst_drop_mask(drop, dseed, (uint32_t)((((long)b * H + h) * S + i) * S + j));
                __bf16 hb = (__bf16)pv;
                Pb[i * SS + j] = __builtin_bit_cast(unsigned short, hb);
            }
        } else {
#pragma unroll
            for (int n = 0; n < ST; ++n) Pb[i * SS + lr + 16 * n] = 0;     // rows S .. SP - 1 (i < SP always: 4 rows per wave pass, SP a multiple of 16)
        }
    }
    __syncthreads();
    // ---- phase C: O = Pd V on the bf16 matrix pipe: Pd rounded to bf16 (phase B), V read with ds_read_b64_tr_b16 (its contraction index
    // is its image's row index), tiles computed transposed so that a lane stores four consecutive columns (as mha_bwd_hb_kernel, round 3)
    float* ob = out ? out + (long)b * S * d + h * HD : nullptr;
    constexpr int PAIR = (NT % 2 == 0) ? 2 : 1, NP = NT / PAIR;
    constexpr int nout = ST * NP;
    const int tq = lr >> 2, tp = lr & 3;
    for (int u = w; u < nout; u += MHA_NW) {
        const int mt = u / NP, n0 = (u - mt * NP) * PAIR * 16;
        f32x4_t acc[PAIR];
#pragma unroll
        for (int q = 0; q < PAIR; ++q) acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < ST; ++ks) {
            const int k0 = ks * 16 + 4 * lq;
            const mha_s16x4_t af = *reinterpret_cast<const mha_s16x4_t*>(Pb + (mt * 16 + lr) * SS + k0);
            mha_s16x4_t bf[PAIR];
            const unsigned short* bp = Vs + (k0 + tq) * HDB + n0 + 4 * tp;
            if constexpr (PAIR == 2) mha_tr_read2(bp, bp + 16, bf[0], bf[PAIR - 1]);
            else bf[0] = mha_tr_read(bp);
#pragma unroll
            for (int q = 0; q < PAIR; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bf[q], af, acc[q], 0, 0, 0);
        }
        const int mrow = mt * 16 + lr;
        if (mrow < S) {
#pragma unroll
            for (int q = 0; q < PAIR; ++q) {
                const int n = n0 + q * 16 + 4 * lq;
                if (ob) *reinterpret_cast<f32x4_t*>(ob + (long)mrow * d + n) = acc[q];
                if (outb) {
                    uint2 pk;
                    { __bf16 h0 = (__bf16)acc[q][0], h1 = (__bf16)acc[q][1], h2 = (__bf16)acc[q][2], h3 = (__bf16)acc[q][3];
                      pk.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
                      pk.y = (uint32_t)__builtin_bit_cast(unsigned short, h2) | ((uint32_t)__builtin_bit_cast(unsigned short, h3) << 16); }
                    *reinterpret_cast<uint2*>(outb + ((long)b * S + mrow) * ldob + h * HD + n) = pk;
                }
            }
        }
    }
}

static int mha_fwd_any(const void* qkv, int qkv_bf16, float* out, float* lse, int B, int S, int H, int hd,
                       float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                       void* out_bf16, long ldob, void* stream);

extern "C" int cst_mha_fwd_b(const float* qkv, float* out, float* lse, int B, int S, int H, int hd,
                             float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                             void* out_bf16, long ldob, void* stream) {
    CST_REQUIRE(out, "cst_mha_fwd: null pointer");
    return mha_fwd_any(qkv, 0, out, lse, B, S, H, hd, drop_p, drop_seed, drop_stream, drop_seed_dev, out_bf16, ldob, stream);
}

extern "C" int cst_mha_fwd_h(const void* qkv_bf16, float* out, float* lse, int B, int S, int H, int hd,
                             float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                             void* out_bf16, long ldob, void* stream) {
    CST_REQUIRE(out || out_bf16, "cst_mha_fwd_h: no output");
    CST_REQUIRE(S <= MHA_SMAX && (hd == 64 || hd == 96), "cst_mha_fwd_h: bf16 qkv is built for S <= %d and head dims 64 / 96 (got S=%d, hd=%d)", MHA_SMAX, S, hd);
    return mha_fwd_any(qkv_bf16, 1, out, lse, B, S, H, hd, drop_p, drop_seed, drop_stream, drop_seed_dev, out_bf16, ldob, stream);
}

static int mha_fwd_any(const void* qkv, int qkv_bf16, float* out, float* lse, int B, int S, int H, int hd,
                       float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                       void* out_bf16, long ldob, void* stream) {
    CST_REQUIRE(!out_bf16 || ldob >= (long)H * hd, "cst_mha_fwd: bf16 leading dimension < d");
    CST_REQUIRE(qkv && lse, "cst_mha_fwd: null pointer");
    CST_REQUIRE(B > 0 && S > 0 && S <= MHA_SMAX_LONG && H > 0, "cst_mha_fwd: S=%d unsupported (max %d)", S, MHA_SMAX_LONG);
    CST_REQUIRE(((uintptr_t)qkv & 15) == 0 && hd % 4 == 0, "cst_mha_fwd: qkv must be 16-byte aligned, hd a multiple of 4");
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)B * H * S * S);
    const float scale = 1.0f / sqrtf((float)hd);
    if (S > MHA_SMAX) {
        const int rc = cst_mha_fwd_long((const float*)qkv, out, lse, B, S, H, hd, scale, dr, out_bf16, ldob, (hipStream_t)stream);
        if (rc != CST_OK) return rc;
        CST_LAUNCH_CHECK("cst_mha_fwd (long)");
        return CST_OK;
    }
    dim3 grid(B * H), block(MHA_NW * 64);
    hipStream_t st = (hipStream_t)stream;
    static const bool hb_off = getenv("CST_MHA_HB_OFF") != nullptr;          // A/B switch: fp32 LDS images for the bf16-input forward
    if (qkv_bf16 && (hd == 64 || hd == 96) && !hb_off) {
        CST_REQUIRE((((uintptr_t)out) & 15) == 0 && (((uintptr_t)out_bf16) & 7) == 0 && (!out_bf16 || ldob % 4 == 0),
                    "cst_mha_fwd_h: out must be 16-byte aligned, its bf16 twin 8-byte aligned with a leading dimension that is a multiple of 4");
        const size_t ldsb = mha_fwd_hb_lds_bytes(S, hd);
#define MHA_HB_LAUNCH(HDV, STV)                                                                                    \
        {                                                                                                          \
            if (ldsb > 64 * 1024) (void)hipFuncSetAttribute((const void*)mha_fwd_hb_kernel<HDV, STV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
            hipLaunchKernelGGL((mha_fwd_hb_kernel<HDV, STV>), grid, block, ldsb, st, (const unsigned short*)qkv, out, lse, S, H, scale, dr,     \
                               (unsigned short*)out_bf16, ldob);                                                   \
        }
#define MHA_HB_CASE(HDV)                                                                                           \
        switch ((S + 15) / 16) {                                                                                   \
            case 1: MHA_HB_LAUNCH(HDV, 1) break;                                                                   \
            case 2: MHA_HB_LAUNCH(HDV, 2) break;                                                                   \
            case 3: MHA_HB_LAUNCH(HDV, 3) break;                                                                   \
            default: MHA_HB_LAUNCH(HDV, 4) break;                                                                  \
        }
        if (hd == 64) { MHA_HB_CASE(64) } else { MHA_HB_CASE(96) }
#undef MHA_HB_CASE
#undef MHA_HB_LAUNCH
        CST_LAUNCH_CHECK("cst_mha_fwd (bf16 images)");
        return CST_OK;
    }
    const size_t lds = sizeof(float) * mha_fwd_lds_floats(S, hd);
    CST_REQUIRE(lds <= 160 * 1024, "cst_mha_fwd: LDS need %zu exceeds 160 KiB", lds);
#define MHA_FWD_LAUNCH(HDV, STV)                                                                                  \
    {                                                                                                             \
        if constexpr (HDV == 64 || HDV == 96) {                                                                    \
            if (qkv_bf16) {                                                                                       \
                if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mha_fwd_kernel<HDV, STV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                hipLaunchKernelGGL((mha_fwd_kernel<HDV, STV, true>), grid, block, lds, st, qkv, out, lse, S, H, scale, dr, (unsigned short*)out_bf16, ldob); \
                break;                                                                                            \
            }                                                                                                     \
        }                                                                                                         \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mha_fwd_kernel<HDV, STV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((mha_fwd_kernel<HDV, STV, false>), grid, block, lds, st, qkv, out, lse, S, H, scale, dr, (unsigned short*)out_bf16, ldob); \
    }
#define MHA_FWD_CASE(HDV)                                                                                         \
    case HDV: {                                                                                                   \
        switch ((S + 15) / 16) {                                                                                  \
            case 1: do MHA_FWD_LAUNCH(HDV, 1) while (0); break;                                                   \
            case 2: do MHA_FWD_LAUNCH(HDV, 2) while (0); break;                                                   \
            case 3: do MHA_FWD_LAUNCH(HDV, 3) while (0); break;                                                   \
            default: do MHA_FWD_LAUNCH(HDV, 4) while (0); break;                                                  \
        }                                                                                                         \
        break;                                                                                                    \
    }
    switch (hd) {
        MHA_FWD_CASE(8) MHA_FWD_CASE(16) MHA_FWD_CASE(32) MHA_FWD_CASE(64) MHA_FWD_CASE(96)
        default: cst_set_error("cst_mha_fwd: head dim %d unsupported (8, 16, 32, 64, 96)", hd); return CST_ERR_ARG;
    }
#undef MHA_FWD_LAUNCH
#undef MHA_FWD_CASE
    CST_LAUNCH_CHECK("cst_mha_fwd");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// MHA backward: recompute P from Q, K and the saved log-sum-exp, then
//   dV = Pd^T dO ; dPd = dO V^T ; dS = P o (dP - rowsum(dP o P)) * scale ; dQ = dS K ; dK = dS^T Q
// with Pd the dropped probabilities.  dqkv [B,S,3d].
//
// All five products run on the exact-fp32 matrix pipe (v_mfma_f32_16x16x4_f32 == an fmaf chain):
// a VALU formulation needs one wave-wide LDS broadcast (1 KiB of LDS return bandwidth) per 4
// FMAs and was LDS-bound at ~650 GB/s of HBM traffic.  One workgroup per (batch row, head), S
// padded to SP = 16*ST rows (pad rows replicate row S-1; P and dS are forced to 0 there):
//   phase A  per 16x16 tile of the S x S plane: Sc = Q K^T and dPd = dO V^T (k split in four
//            contiguous segments, one per 16-lane group, so operands are b128 row reads of the
//            padded [SP][HD+4] images); P, dP, the tile's row sums of dP o P; Pd -> LDS
//   phase B  delta = sum of the row-sum partials in tile order (deterministic); dS -> LDS
//   phase C  the 3 * ST * ceil(HD/16) output tiles of dQ, dK, dV, contraction over the S x S images
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline size_t mha_bwd_lds_floats(int S, int hd) {
    const size_t SP = (size_t)((S + 15) / 16) * 16;
    return 4 * SP * (hd + 4) + 16 + 2 * SP * (SP + 4) + SP * 4 + SP;
}

template <int HD, int ST, bool QB = false>
__global__ __launch_bounds__(MHA_NW * 64) void mha_bwd_kernel(const void* __restrict__ qkv, const void* __restrict__ dout,
                                                      const float* __restrict__ lse, float* __restrict__ dqkv,
                                                      int S, int H, float scale, CstDrop drop,
                                                      unsigned short* __restrict__ dqkvb, long lddb) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int HD4 = HD / 4, HDS = HD + 4, SEG = HD / 4, NT = (HD + 15) / 16;
    constexpr int SP = ST * 16, SS = SP + 4, KSEG = SP / 4;     // ST = ceil(S / 16): every loop below unrolls
    float* Qs = smem;                    // [SP][HDS]
    float* Ks = Qs + SP * HDS;
    float* Vs = Ks + SP * HDS;
    float* Os = Vs + SP * HDS;           // dO
    float* Pm = Os + SP * HDS + 16;      // [SP][SS]  Pd[i][j]   (16 floats of slack: HD < 16 reads past a row end)
    float* Dm = Pm + SP * SS;            // [SP][SS]  dS[i][j]
    float* part = Dm + SP * SS;          // [SP][4]   row sums of dP o P per column tile
    float* lse_s = part + SP * 4;        // [SP]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int d = H * HD;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const long base = (long)b * S * 3 * d + h * HD;          // element offsets of this (b, h) slice in qkv / dout
    const long dob = (long)b * S * d + h * HD;

    // stage Q, K, V, dO: every 16-byte load is issued before the first LDS store
    {
        constexpr int nel = SP * HD4;
        constexpr int NTHR = MHA_NW * 64, QIT = (nel + NTHR - 1) / NTHR;
        float4 tq[QIT], tk[QIT], tv[QIT], to[QIT];
#pragma unroll
        for (int it = 0; it < QIT; ++it) {
            // unconditional (clamped) loads: iterations past the tile re-read its last element
            const int e = min((int)threadIdx.x + NTHR * it, nel - 1);
            const int i = min(e / HD4, S - 1), c = (e % HD4) * 4;
            const long src = base + (long)i * 3 * d + c;
            tq[it] = mha_ld4<QB>(qkv, src);
            tk[it] = mha_ld4<QB>(qkv, src + d);
            tv[it] = mha_ld4<QB>(qkv, src + 2 * d);
            to[it] = mha_ld4<QB>(dout, dob + (long)i * d + c);
        }
        if (threadIdx.x < SP) lse_s[threadIdx.x] = (int)threadIdx.x < S ? lse[((long)b * H + h) * S + threadIdx.x] : 0.f;
#pragma unroll
        for (int it = 0; it < QIT; ++it) {
            const int e = threadIdx.x + NTHR * it;
            if (e < nel) {
                const int o = (e / HD4) * HDS + (e % HD4) * 4;
                *reinterpret_cast<float4*>(&Qs[o]) = tq[it];
                *reinterpret_cast<float4*>(&Ks[o]) = tk[it];
                *reinterpret_cast<float4*>(&Vs[o]) = tv[it];
                *reinterpret_cast<float4*>(&Os[o]) = to[it];
            }
        }
    }
    __syncthreads();

    // ---- phase A -----------------------------------------------------------------------------
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    constexpr int ntile = ST * ST, TPW = (ntile + MHA_NW - 1) / MHA_NW;
    f32x4_t Pf[TPW], Df[TPW];            // this wave's tiles, kept for phase B
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int t = w + MHA_NW * tt;
        Pf[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        Df[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (t < ntile) {
            const int mt = t / ST, nt = t % ST;
            const float* qa = Qs + (mt * 16 + lr) * HDS + lq * SEG;
            const float* oa = Os + (mt * 16 + lr) * HDS + lq * SEG;
            const float* ka = Ks + (nt * 16 + lr) * HDS + lq * SEG;
            const float* va = Vs + (nt * 16 + lr) * HDS + lq * SEG;
            f32x4_t accS = {0.f, 0.f, 0.f, 0.f}, accD = {0.f, 0.f, 0.f, 0.f};
            if constexpr (SEG % 4 == 0) {
#pragma unroll
                for (int k4 = 0; k4 < SEG / 4; ++k4) {
                    const float4 q4 = *reinterpret_cast<const float4*>(qa + k4 * 4);
                    const float4 kx = *reinterpret_cast<const float4*>(ka + k4 * 4);
                    const float4 o4 = *reinterpret_cast<const float4*>(oa + k4 * 4);
                    const float4 v4 = *reinterpret_cast<const float4*>(va + k4 * 4);
                    accS = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.x, kx.x, accS, 0, 0, 0);
                    accD = __builtin_amdgcn_mfma_f32_16x16x4f32(o4.x, v4.x, accD, 0, 0, 0);
                    accS = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.y, kx.y, accS, 0, 0, 0);
                    accD = __builtin_amdgcn_mfma_f32_16x16x4f32(o4.y, v4.y, accD, 0, 0, 0);
                    accS = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.z, kx.z, accS, 0, 0, 0);
                    accD = __builtin_amdgcn_mfma_f32_16x16x4f32(o4.z, v4.z, accD, 0, 0, 0);
                    accS = __builtin_amdgcn_mfma_f32_16x16x4f32(q4.w, kx.w, accS, 0, 0, 0);
                    accD = __builtin_amdgcn_mfma_f32_16x16x4f32(o4.w, v4.w, accD, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int k = 0; k < SEG; ++k) {
                    accS = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[k], ka[k], accS, 0, 0, 0);
                    accD = __builtin_amdgcn_mfma_f32_16x16x4f32(oa[k], va[k], accD, 0, 0, 0);
                }
            }
            const int j = nt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = mt * 16 + lq * 4 + r;
                const bool valid = i < S && j < S;
                const float pv = valid ? expf(accS[r] * scale - lse_s[i]) : 0.f;
                float mask = 1.f;
                if (drop.p > 0.f && valid)
                    mask = cst_drop_mask(drop, dseed, (uint32_t)((((long)b * H + h) * S + i) * S + j));
                const float dp = valid ? accD[r] * mask : 0.f;
                Pm[i * SS + j] = pv * mask;
                float rs = dp * pv;                       // row sum over this tile's 16 columns (one 16-lane group)
                rs = row16_sum(rs);
                if (lr == 0) part[i * 4 + nt] = rs;
                Pf[tt][r] = pv;
                Df[tt][r] = dp;
            }
        }
    }
    __syncthreads();
    // ---- phase B -----------------------------------------------------------------------------
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int t = w + MHA_NW * tt;
        if (t < ntile) {
            const int mt = t / ST, nt = t % ST;
            const int j = nt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = mt * 16 + lq * 4 + r;
                float delta = 0.f;
#pragma unroll
                for (int n = 0; n < ST; ++n) delta += part[i * 4 + n];
                Dm[i * SS + j] = Pf[tt][r] * (Df[tt][r] - delta) * scale;
            }
        }
    }
    __syncthreads();
    // ---- phase C -----------------------------------------------------------------------------
    // Each work item is one 16-row block of dQ, dK or dV times PAIR adjacent 16-column tiles: the two
    // accumulator chains share the A operand and hide each other's MFMA latency.
    float* dq = dqkv + (long)b * S * 3 * d + h * HD;
    constexpr int PAIR = (NT % 2 == 0) ? 2 : 1, NP = NT / PAIR;
    constexpr int per = ST * NP, nout = 3 * per;
    for (int u = w; u < nout; u += MHA_NW) {
        const int which = u / per, rem = u - which * per;
        const int mt = rem / NP, n0 = (rem - mt * NP) * PAIR * 16;
        // which 0: dQ[i][c] = sum_j dS[i][j] K[j][c]   A[m=i][k=j] = Dm[i][j]  (k walks a row)
        // which 1: dK[j][c] = sum_i dS[i][j] Q[i][c]   A[m=j][k=i] = Dm[i][j]  (k walks a column)
        // which 2: dV[j][c] = sum_i Pd[i][j] dO[i][c]
        const float* a = which == 0 ? Dm + (mt * 16 + lr) * SS + lq * KSEG
                                    : (which == 1 ? Dm : Pm) + (lq * KSEG) * SS + mt * 16 + lr;
        const int as = which == 0 ? 1 : SS;
        const float* bp = (which == 0 ? Ks : (which == 1 ? Qs : Os)) + (lq * KSEG) * HDS + n0 + lr;
        f32x4_t acc[PAIR];
#pragma unroll
        for (int q = 0; q < PAIR; ++q) acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < KSEG; ++k) {
            const float av = a[k * as];
#pragma unroll
            for (int q = 0; q < PAIR; ++q)
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bp[k * HDS + q * 16], acc[q], 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < PAIR; ++q) {
            const int n = n0 + q * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mt * 16 + lq * 4 + r;
                if (m < S && n < HD) {
                    if (dqkv) dq[(long)m * 3 * d + which * d + n] = acc[q][r];
                    if (dqkvb) {                                  // A operand of the in-projection dgrad / weight gradient
                        __bf16 hh = (__bf16)acc[q][r];
                        dqkvb[((long)b * S + m) * lddb + which * d + h * HD + n] = __builtin_bit_cast(unsigned short, hh);
                    }
                }
            }
        }
    }
}

extern "C" int cst_mha_bwd_b(const float* qkv, const float* dout, const float* lse, float* dqkv,
                             int B, int S, int H, int hd,
                             float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                             void* dqkv_bf16, long lddb, void* stream);
extern "C" int cst_mha_bwd(const float* qkv, const float* dout, const float* lse, float* dqkv,
                           int B, int S, int H, int hd,
                           float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                           void* stream) {
    return cst_mha_bwd_b(qkv, dout, lse, dqkv, B, S, H, hd, drop_p, drop_seed, drop_stream, drop_seed_dev, nullptr, 0, stream);
}

// ---------------------------------------------------------------------------------------------
// MHA backward, bf16 I/O (cst_mha_bwd_h), HD a multiple of 32: the LDS images of Q, K, V, dO stay bf16.
//
// Why (profiles/round2_kernel_stats: mha_bwd_kernel<96, 3, true> 99.5 us x 12 per step, ~22 TFLOP/s): with fp32 images the kernel
// needs 97 KiB of LDS at S = 36, hd = 96 -- ONE workgroup per CU, so nothing hides a workgroup's load phase (27 KB of strided rows,
// one memory round trip) or its write-out.  bf16 images are 40 KiB (61 KiB in all): two workgroups per CU, one loading while the
// other computes.  Phase A (Q K^T and dO V^T) then runs on v_mfma_f32_16x16x32_bf16 straight from the images: the products of
// bf16 values are exact in fp32 either way, only the summation order differs from the fp32-image kernel (results agree to a few
// ulps, not bit for bit).  Phase C keeps the fp32 matrix pipe: its A operand (dS, Pd) is fp32 and is not rounded; its B operand
// (K, Q, dO) is widened on the fly from the bf16 image (a shift).
//   images: [SP][HD + 8] bf16 (208-byte rows at hd = 96: 16-byte aligned, 13 x 16 B so the 16 rows of a fragment read spread over
//   the banks)
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline size_t mha_bwd_hb_lds_bytes(int S, int hd) {
    const size_t SP = (size_t)((S + 15) / 16) * 16;
    return 4 * SP * (hd + 8) * 2 + 64 + 2 * SP * (SP + 4) * 2 + sizeof(float) * (SP * 4 + SP);
}

template <int HD, int ST>
__global__ __launch_bounds__(MHA_NW * 64) void mha_bwd_hb_kernel(const unsigned short* __restrict__ qkv, const unsigned short* __restrict__ dout,
                                                         const float* __restrict__ lse, float* __restrict__ dqkv,
                                                         int S, int H, float scale, CstDrop drop,
                                                         unsigned short* __restrict__ dqkvb, long lddb) {
    static_assert(HD % 32 == 0, "bf16 MFMA k-steps of 32");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int HD4 = HD / 4, HDB = HD + 8, NT = HD / 16;
    constexpr int SP = ST * 16, SS = SP + 4;
    unsigned short* Qs = reinterpret_cast<unsigned short*>(smem_raw);      // [SP][HDB]
    unsigned short* Ks = Qs + SP * HDB;
    unsigned short* Vs = Ks + SP * HDB;
    unsigned short* Os = Vs + SP * HDB;                                     // dO
    unsigned short* Pm = Os + SP * HDB + 32;                                // [SP][SS]  bf16 Pd[i][j]
    unsigned short* Dm = Pm + SP * SS;                                      // [SP][SS]  bf16 dS[i][j]
    float* part = reinterpret_cast<float*>(Dm + SP * SS);                   // [SP][4]
    float* lse_s = part + SP * 4;                                           // [SP]
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int d = H * HD;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const long base = (long)b * S * 3 * d + h * HD;
    const long dob = (long)b * S * d + h * HD;

    // stage Q, K, V, dO (4 bf16 = 8 bytes per load): every load is issued before the first LDS store
    {
        constexpr int nel = SP * HD4;
        constexpr int NTHR = MHA_NW * 64, QIT = (nel + NTHR - 1) / NTHR;
        uint2 tq[QIT], tk[QIT], tv[QIT], to[QIT];
#pragma unroll
        for (int it = 0; it < QIT; ++it) {
            const int e = min((int)threadIdx.x + NTHR * it, nel - 1);
            const int i = min(e / HD4, S - 1), c = (e % HD4) * 4;
            const long src = base + (long)i * 3 * d + c;
            tq[it] = *reinterpret_cast<const uint2*>(qkv + src);
            tk[it] = *reinterpret_cast<const uint2*>(qkv + src + d);
            tv[it] = *reinterpret_cast<const uint2*>(qkv + src + 2 * d);
            to[it] = *reinterpret_cast<const uint2*>(dout + dob + (long)i * d + c);
        }
        if (threadIdx.x < SP) lse_s[threadIdx.x] = (int)threadIdx.x < S ? lse[((long)b * H + h) * S + threadIdx.x] : 0.f;
#pragma unroll
        for (int it = 0; it < QIT; ++it) {
            const int e = threadIdx.x + NTHR * it;
            if (e < nel) {
                const int o = (e / HD4) * HDB + (e % HD4) * 4;
                *reinterpret_cast<uint2*>(&Qs[o]) = tq[it];
                *reinterpret_cast<uint2*>(&Ks[o]) = tk[it];
                *reinterpret_cast<uint2*>(&Vs[o]) = tv[it];
                *reinterpret_cast<uint2*>(&Os[o]) = to[it];
            }
        }
    }
    __syncthreads();

    // ---- phase A: S = Q K^T and dP = dO V^T per 16 x 16 tile, HD / 32 bf16 MFMAs each ---------
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    constexpr int ntile = ST * ST, TPW = (ntile + MHA_NW - 1) / MHA_NW;
    f32x4_t Pf[TPW], Df[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int t = w + MHA_NW * tt;
        Pf[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        Df[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (t < ntile) {
            const int mt = t / ST, nt = t % ST;
            const unsigned short* qa = Qs + (mt * 16 + lr) * HDB + lq * 8;
            const unsigned short* oa = Os + (mt * 16 + lr) * HDB + lq * 8;
            const unsigned short* ka = Ks + (nt * 16 + lr) * HDB + lq * 8;
            const unsigned short* va = Vs + (nt * 16 + lr) * HDB + lq * 8;
            f32x4_t accS = {0.f, 0.f, 0.f, 0.f}, accD = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < HD / 32; ++ks) {
                const mha_bf16x8_t q8 = *reinterpret_cast<const mha_bf16x8_t*>(qa + ks * 32);
                const mha_bf16x8_t k8 = *reinterpret_cast<const mha_bf16x8_t*>(ka + ks * 32);
                const mha_bf16x8_t o8 = *reinterpret_cast<const mha_bf16x8_t*>(oa + ks * 32);
                const mha_bf16x8_t v8 = *reinterpret_cast<const mha_bf16x8_t*>(va + ks * 32);
                accS = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q8, k8, accS, 0, 0, 0);
                accD = __builtin_amdgcn_mfma_f32_16x16x32_bf16(o8, v8, accD, 0, 0, 0);
            }
            const int j = nt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = mt * 16 + lq * 4 + r;
                const bool valid = i < S && j < S;
                const float pv = valid ? expf(accS[r] * scale - lse_s[i]) : 0.f;
                float mask = 1.f;
                if (drop.p > 0.f && valid)
                    mask = cst_drop_mask(drop, dseed, (uint32_t)((((long)b * H + h) * S + i) * S + j));
                const float dp = valid ? accD[r] * mask : 0.f;
                { __bf16 hb = (__bf16)(pv * mask); Pm[i * SS + j] = __builtin_bit_cast(unsigned short, hb); }
                float rs = dp * pv;
                rs = row16_sum(rs);
                if (lr == 0) part[i * 4 + nt] = rs;
                Pf[tt][r] = pv;
                Df[tt][r] = dp;
            }
        }
    }
    __syncthreads();
    // ---- phase B -----------------------------------------------------------------------------
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int t = w + MHA_NW * tt;
        if (t < ntile) {
            const int mt = t / ST, nt = t % ST;
            const int j = nt * 16 + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = mt * 16 + lq * 4 + r;
                float delta = 0.f;
#pragma unroll
                for (int n = 0; n < ST; ++n) delta += part[i * 4 + n];
                __bf16 hb = (__bf16)(Pf[tt][r] * (Df[tt][r] - delta) * scale);
                Dm[i * SS + j] = __builtin_bit_cast(unsigned short, hb);
            }
        }
    }
    __syncthreads();
    // ---- phase C: dQ = dS K, dK = dS^T Q, dV = Pd^T dO on the bf16 matrix pipe (v_mfma_f32_16x16x16_bf16, contraction over the sequence
    // in steps of 16).  dS and Pd are rounded to bf16 once (phase A / B wrote them that way); every operand whose contraction index is its
    // image's ROW index -- K, Q, dO always, dS and Pd for the two transposed products -- comes from the hardware transposed read, so no
    // transposed image exists.  (Round 2 ran these on the fp32 pipe from scalar LDS reads: 12 MFMAs and 24 reads per tile where this is 3 and 6.)
    float* dq = dqkv ? dqkv + (long)b * S * 3 * d + h * HD : nullptr;
    constexpr int PAIR = (NT % 2 == 0) ? 2 : 1, NP = NT / PAIR;
    constexpr int per = ST * NP, nout = 3 * per;
    const int tq = lr >> 2, tp = lr & 3;                  // this lane's (row, column group) inside a transposed-read block
    for (int u = w; u < nout; u += MHA_NW) {
        const int which = u / per, rem = u - which * per;
        const int mt = rem / NP, n0 = (rem - mt * NP) * PAIR * 16;
        const unsigned short* aimg = which == 2 ? Pm : Dm;
        const unsigned short* bimg = which == 0 ? Ks : (which == 1 ? Qs : Os);
        f32x4_t acc[PAIR];
#pragma unroll
        for (int q = 0; q < PAIR; ++q) acc[q] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < ST; ++ks) {
            const int k0 = ks * 16 + 4 * lq;              // this 16-lane group's four contraction indices
            mha_s16x4_t af, bf[PAIR];
            const unsigned short* bp = bimg + (k0 + tq) * HDB + n0 + 4 * tp;
            const unsigned short* ap = aimg + (k0 + tq) * SS + mt * 16 + 4 * tp;                               // dS^T / Pd^T
            if (which == 0) {
                af = *reinterpret_cast<const mha_s16x4_t*>(aimg + (mt * 16 + lr) * SS + k0);                    // dS[i][j .. j + 3]: a row read
                if constexpr (PAIR == 2) mha_tr_read2(bp, bp + 16, bf[0], bf[PAIR - 1]);
                else bf[0] = mha_tr_read(bp);
            } else {
                if constexpr (PAIR == 2) mha_tr_read3(ap, bp, bp + 16, af, bf[0], bf[PAIR - 1]);
                else mha_tr_read2(ap, bp, af, bf[0]);
            }
#pragma unroll
            for (int q = 0; q < PAIR; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(bf[q], af, acc[q], 0, 0, 0);   // operands swapped: the tile comes out transposed
        }
        // transposed tile: this lane holds columns n .. n + 3 of output row m -- one 16-byte (fp32) / 8-byte (bf16) store each
        const int m = mt * 16 + lr;
        if (m < S) {
#pragma unroll
            for (int q = 0; q < PAIR; ++q) {
                const int n = n0 + q * 16 + 4 * lq;
                if (dq) *reinterpret_cast<f32x4_t*>(dq + (long)m * 3 * d + which * d + n) = acc[q];
                if (dqkvb) {
                    uint2 pk;
                    { __bf16 h0 = (__bf16)acc[q][0], h1 = (__bf16)acc[q][1], h2 = (__bf16)acc[q][2], h3 = (__bf16)acc[q][3];
                      pk.x = (uint32_t)__builtin_bit_cast(unsigned short, h0) | ((uint32_t)__builtin_bit_cast(unsigned short, h1) << 16);
                      pk.y = (uint32_t)__builtin_bit_cast(unsigned short, h2) | ((uint32_t)__builtin_bit_cast(unsigned short, h3) << 16); }
                    *reinterpret_cast<uint2*>(dqkvb + ((long)b * S + m) * lddb + which * d + h * HD + n) = pk;
                }
            }
        }
    }
}


static int mha_bwd_any(const void* qkv, const void* dout, int io_bf16, const float* lse, float* dqkv, int B, int S, int H, int hd,
                       float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                       void* dqkv_bf16, long lddb, void* stream);

extern "C" int cst_mha_bwd_b(const float* qkv, const float* dout, const float* lse, float* dqkv,
                             int B, int S, int H, int hd,
                             float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                             void* dqkv_bf16, long lddb, void* stream) {
    CST_REQUIRE(dqkv, "cst_mha_bwd: null pointer");
    return mha_bwd_any(qkv, dout, 0, lse, dqkv, B, S, H, hd, drop_p, drop_seed, drop_stream, drop_seed_dev, dqkv_bf16, lddb, stream);
}

extern "C" int cst_mha_bwd_h(const void* qkv_bf16, const void* dout_bf16, const float* lse, float* dqkv,
                             int B, int S, int H, int hd,
                             float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                             void* dqkv_bf16, long lddb, void* stream) {
    CST_REQUIRE(dqkv || dqkv_bf16, "cst_mha_bwd_h: no output");
    CST_REQUIRE(S <= MHA_SMAX && (hd == 64 || hd == 96), "cst_mha_bwd_h: bf16 qkv / dout are built for S <= %d and head dims 64 / 96 (got S=%d, hd=%d)", MHA_SMAX, S, hd);
    return mha_bwd_any(qkv_bf16, dout_bf16, 1, lse, dqkv, B, S, H, hd, drop_p, drop_seed, drop_stream, drop_seed_dev, dqkv_bf16, lddb, stream);
}

static int mha_bwd_any(const void* qkv, const void* dout, int io_bf16, const float* lse, float* dqkv, int B, int S, int H, int hd,
                       float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                       void* dqkv_bf16, long lddb, void* stream) {
    CST_REQUIRE(!dqkv_bf16 || lddb >= 3L * H * hd, "cst_mha_bwd: bf16 leading dimension < 3d");
    CST_REQUIRE(qkv && dout && lse, "cst_mha_bwd: null pointer");
    CST_REQUIRE(B > 0 && S > 0 && S <= MHA_SMAX_LONG && H > 0, "cst_mha_bwd: S=%d unsupported (max %d)", S, MHA_SMAX_LONG);
    CST_REQUIRE((((uintptr_t)qkv | (uintptr_t)dout) & 15) == 0 && hd % 4 == 0, "cst_mha_bwd: qkv / dout must be 16-byte aligned, hd a multiple of 4");
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)B * H * S * S);
    const float scale = 1.0f / sqrtf((float)hd);
    if (S > MHA_SMAX) {
        const int rc = cst_mha_bwd_long((const float*)qkv, (const float*)dout, lse, dqkv, B, S, H, hd, scale, dr, dqkv_bf16, lddb, (hipStream_t)stream);
        if (rc != CST_OK) return rc;
        CST_LAUNCH_CHECK("cst_mha_bwd (long)");
        return CST_OK;
    }
    dim3 grid(B * H), block(MHA_NW * 64);
    hipStream_t st = (hipStream_t)stream;
    static const bool hb_off = getenv("CST_MHA_HB_OFF") != nullptr;          // A/B switch: fp32 LDS images for the bf16-I/O backward
    if (io_bf16 && (hd == 64 || hd == 96) && !hb_off) {
        // bf16 LDS images: two workgroups per CU (see mha_bwd_hb_kernel)
        CST_REQUIRE((((uintptr_t)dqkv) & 15) == 0 && (((uintptr_t)dqkv_bf16) & 7) == 0 && (!dqkv_bf16 || lddb % 4 == 0),
                    "cst_mha_bwd_h: dqkv must be 16-byte aligned, its bf16 twin 8-byte aligned with a leading dimension that is a multiple of 4");
        const size_t ldsb = mha_bwd_hb_lds_bytes(S, hd);
#define MHA_HB_LAUNCH(HDV, STV)                                                                                    \
        {                                                                                                          \
            if (ldsb > 64 * 1024) (void)hipFuncSetAttribute((const void*)mha_bwd_hb_kernel<HDV, STV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb); \
            hipLaunchKernelGGL((mha_bwd_hb_kernel<HDV, STV>), grid, block, ldsb, st, (const unsigned short*)qkv, (const unsigned short*)dout, lse, dqkv, S, H, \
                               scale, dr, (unsigned short*)dqkv_bf16, lddb);                                       \
        }
#define MHA_HB_CASE(HDV)                                                                                           \
        switch ((S + 15) / 16) {                                                                                   \
            case 1: MHA_HB_LAUNCH(HDV, 1) break;                                                                   \
            case 2: MHA_HB_LAUNCH(HDV, 2) break;                                                                   \
            case 3: MHA_HB_LAUNCH(HDV, 3) break;                                                                   \
            default: MHA_HB_LAUNCH(HDV, 4) break;                                                                  \
        }
        if (hd == 64) { MHA_HB_CASE(64) } else { MHA_HB_CASE(96) }
#undef MHA_HB_CASE
#undef MHA_HB_LAUNCH
        CST_LAUNCH_CHECK("cst_mha_bwd (bf16 images)");
        return CST_OK;
    }
    const size_t lds = sizeof(float) * mha_bwd_lds_floats(S, hd);
    CST_REQUIRE(lds <= 160 * 1024, "cst_mha_bwd: LDS need %zu exceeds 160 KiB", lds);
#define MHA_BWD_LAUNCH(HDV, STV)                                                                                  \
    {                                                                                                             \
        if constexpr (HDV == 64 || HDV == 96) {                                                                    \
            if (io_bf16) {                                                                                        \
                if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mha_bwd_kernel<HDV, STV, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
                hipLaunchKernelGGL((mha_bwd_kernel<HDV, STV, true>), grid, block, lds, st, qkv, dout, lse, dqkv, S, H, scale, dr, (unsigned short*)dqkv_bf16, lddb); \
                break;                                                                                            \
            }                                                                                                     \
        }                                                                                                         \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)mha_bwd_kernel<HDV, STV, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((mha_bwd_kernel<HDV, STV, false>), grid, block, lds, st, qkv, dout, lse, dqkv, S, H, scale, dr, (unsigned short*)dqkv_bf16, lddb); \
    }
#define MHA_BWD_CASE(HDV)                                                                                         \
    case HDV: {                                                                                                   \
        switch ((S + 15) / 16) {                                                                                  \
            case 1: do MHA_BWD_LAUNCH(HDV, 1) while (0); break;                                                   \
            case 2: do MHA_BWD_LAUNCH(HDV, 2) while (0); break;                                                   \
            case 3: do MHA_BWD_LAUNCH(HDV, 3) while (0); break;                                                   \
            default: do MHA_BWD_LAUNCH(HDV, 4) while (0); break;                                                  \
        }                                                                                                         \
        break;                                                                                                    \
    }
    switch (hd) {
        MHA_BWD_CASE(8) MHA_BWD_CASE(16) MHA_BWD_CASE(32) MHA_BWD_CASE(64) MHA_BWD_CASE(96)
        default: cst_set_error("cst_mha_bwd: head dim %d unsupported (8, 16, 32, 64, 96)", hd); return CST_ERR_ARG;
    }
#undef MHA_BWD_LAUNCH
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
// global -> LDS copy of n4 float4 with four loads in flight per thread (a run-time-bounded one-load
// loop waits out a full memory round trip per iteration)
__device__ __forceinline__ void stage_f4(const float4* __restrict__ src, float4* dst, int n4) {
    const int bd = blockDim.x;
    for (int base = 0; base < n4; base += 4 * bd) {
        float4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = src[min(base + u * bd + (int)threadIdx.x, n4 - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = base + u * bd + threadIdx.x;
            if (e < n4) dst[e] = t[u];
        }
    }
}

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
    const float qv = qb[min((int)threadIdx.x, D - 1)];             // in flight together with the tile
    stage_f4(mb4, reinterpret_cast<float4*>(ms), n4);
    if (threadIdx.x < D) qs[threadIdx.x] = qv;
    for (int c = threadIdx.x + blockDim.x; c < D; c += blockDim.x) qs[c] = qb[c];
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
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)B * 2 * D);
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
    const float gv = gb[min((int)threadIdx.x, D - 1)];             // in flight together with the tile
    const float pv = p[(long)b * L + min((int)threadIdx.x, L - 1)];
    stage_f4(mb4, reinterpret_cast<float4*>(ms), n4);
    if (threadIdx.x < D) gs[threadIdx.x] = gv;
    for (int c = threadIdx.x + blockDim.x; c < D; c += blockDim.x) gs[c] = gb[c];
    if (threadIdx.x < L) ps[threadIdx.x] = pv;
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
        float* o = dq + (long)b * lddq + c;
        const float prev = dq_accumulate ? *o : 0.f;
        float a = 0.f;
        for (int j0 = 0; j0 < L; j0 += 8) {               // dmem += ...: eight read-modify-writes in flight
            float old[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) old[u] = dmb[(long)min(j0 + u, L - 1) * D + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                if (j < L) {
                    a += ds[j] * ms[j * D + c];
                    dmb[(long)j * D + c] = old[u] + ps[j] * g + ds[j] * qc;
                }
            }
        }
        *o = prev + a;
    }
}

// ---------------------------------------------------------------------------------------------
// Decoder attention backward for ALL T decode steps of a batch row in one workgroup (teacher-forced /
// free-running decodes, where the FFN gradients of every step are known before the recurrence runs):
// the memory tile is staged once instead of T times, d memory is accumulated in LDS and written once,
// and the dropout of the FFN input (rnn.py:78-79, call-site stream drop.stream + s, element b*2D + c)
// is applied on the way in.  g = diffn[b, s, 0:2D] holds d[h | a] of the FFN input; on return
// diffn[b, s, 0:D] = dropout(g)[0:D] + dq (the total d h_s that the LSTM cell backward consumes).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void dot_attn_bwd_steps_kernel(float* __restrict__ diffn, long ldrow, long gstep,
                                                                 const float* __restrict__ q, long ldq, long qstep,
                                                                 const float* __restrict__ mem, const float* __restrict__ p,
                                                                 float* __restrict__ dmem, int B, int T, int L, int D, float scale,
                                                                 CstDrop drop) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* ms = dsm;                 // [L][D] memory tile
    float* dms = ms + L * D;         // [L][D] d memory accumulated over the steps
    float* gs = dms + L * D;         // [D]    dropped d a_s
    float* qs = gs + D;              // [D]    h_s
    float* ds = qs + D;              // [64]
    float* ps = ds + 64;             // [64]
    const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int tid = threadIdx.x;
    stage_f4(reinterpret_cast<const float4*>(mem + (long)b * L * D), reinterpret_cast<float4*>(ms), L * D / 4);
    for (int e = tid; e < L * D; e += blockDim.x) dms[e] = 0.f;
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    float* gb = diffn + (long)b * ldrow;
    const float* qb = q + (long)b * ldq;
    // step 0 operands; inside the loop the next step's are requested before this step's arithmetic
    float graw = tid < 2 * D ? gb[tid] : 0.f;
    float qv = tid < D ? qb[tid] : 0.f;
    float pv = tid < L ? p[(long)b * L + tid] : 0.f;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        float gh = 0.f;
        if (tid < 2 * D) {
            float gd = graw;
            if (drop.p > 0.f)
                gd *= ((cst_mix32(dseed, drop.stream + (uint32_t)s, (uint32_t)((long)b * 2 * D + tid) + drop.base) >> 8) >= drop.thresh) ? drop.scale : 0.0f;
            if (tid >= D) gs[tid - D] = gd; else gh = gd;
        }
        if (tid < D) qs[tid] = qv;
        if (tid < L) ps[tid] = pv;
        if (s + 1 < T) {                                   // prefetch
            graw = tid < 2 * D ? gb[(long)(s + 1) * gstep + tid] : 0.f;
            qv = tid < D ? qb[(long)(s + 1) * qstep + tid] : 0.f;
            pv = tid < L ? p[((long)(s + 1) * B + b) * L + tid] : 0.f;
        }
        __syncthreads();
        for (int j = w; j < L; j += nw) {
            float a = 0.f;
            for (int c = lane; c < D; c += 64) a += gs[c] * ms[j * D + c];
            a = wave_sum(a);
            if (lane == 0) ds[j] = a;                      // dp_j
        }
        __syncthreads();
        float delta = 0.f;
        for (int j = 0; j < L; ++j) delta += ds[j] * ps[j];
        __syncthreads();
        if (tid < L) ds[tid] = ps[tid] * (ds[tid] - delta) * scale;
        __syncthreads();
        if (tid < D) {
            const float g = gs[tid], qc = qs[tid];
            float a = 0.f;
            for (int j = 0; j < L; ++j) {
                const float dsj = ds[j];
                a += dsj * ms[j * D + tid];
                dms[j * D + tid] += ps[j] * g + dsj * qc;
            }
            gb[(long)s * gstep + tid] = gh + a;
        }
        __syncthreads();
    }
    float* dmb = dmem + (long)b * L * D;
    for (int e = tid; e < L * D; e += blockDim.x) dmb[e] += dms[e];
}

extern "C" int cst_dot_attn_bwd_steps(float* diffn, long ldrow, long gstep, const float* q, long ldq, long qstep,
                                      const float* mem, const float* p, float* dmem, int B, int T, int L, int D,
                                      float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream) {
    CST_REQUIRE(diffn && q && mem && p && dmem, "cst_dot_attn_bwd_steps: null pointer");
    CST_REQUIRE(B > 0 && T > 0 && L > 0 && L <= MHA_SMAX && D > 0 && D % 4 == 0 && 2 * D <= 1024,
                "cst_dot_attn_bwd_steps: L=%d (max %d) or D=%d (max 512) unsupported", L, MHA_SMAX, D);
    CST_REQUIRE(((uintptr_t)mem & 15) == 0, "cst_dot_attn_bwd_steps: mem must be 16-byte aligned");
    const size_t lds = sizeof(float) * ((size_t)2 * L * D + 2 * D + 128);
    CST_REQUIRE(lds <= 160 * 1024, "cst_dot_attn_bwd_steps: tiles of %zu bytes exceed the 160 KiB LDS", lds);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)dot_attn_bwd_steps_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)B * 2 * D);
    hipLaunchKernelGGL(dot_attn_bwd_steps_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, diffn, ldrow, gstep, q, ldq, qstep, mem, p,
                       dmem, B, T, L, D, 1.0f / sqrtf((float)D), dr);
    CST_LAUNCH_CHECK("cst_dot_attn_bwd_steps");
    return CST_OK;
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
