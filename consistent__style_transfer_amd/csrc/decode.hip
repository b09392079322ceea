// Decoder-step kernels for gfx950: the attention-LSTM decode loop of DenoiseLSTM.forward (reference src/model/rnn.py:72-97) as FOUR
// dependent launches per step instead of six, none of them a split-K reduce:
//
//   cst_dec_gates        token choice + embedding (+ dropout) of x_t, gates = [x_t | h_{t-1}] [W_ih | W_hh]^T + b, LSTM cell   (rnn.py:74, :88-96)
//   cst_dot_attn_fwd     single-query attention of h_t over the encoder states, dropout of [h_t | a_t]                          (rnn.py:76-79, attention.hip)
//   cst_gemm_bf16_skinny fn_1 + LeakyReLU                                                                                       (rnn.py:79-80)
//   cst_gemm_bf16_argmax fn_2 into the stacked (B, T, V) output; its epilogue also leaves each row's arg-max                    (rnn.py:80, :83-92)
//
// Why this shape.  The step is a chain of M = batch = 256 products whose kernels are bound by what they wait for, not by arithmetic
// (profiles/round3_decode_step.txt): a launch costs ~3 us before its first useful cycle, a K-tile fetched one tile ahead of a
// 0.25 us MFMA block costs a full L2 round trip, and a split-K product pays a second launch for its reduce.  So:
//   * the small products take their WHOLE K into LDS at once -- every DMA piece of both operands is requested before the first wait
//     (32 x 64 / 32 x 32 output tiles: 120-128 KiB of LDS, one workgroup per CU, 128-256 workgroups) -- one exposed round trip per
//     launch instead of one per K-tile, and no K split, hence no slab round trip and no reduce launch;
//   * the gate columns of a tile are PERMUTED: tile column 16 q + j is gate q of hidden unit 16 n + j, so a workgroup holds all four
//     gate pre-activations of its 32 rows x 16 units and finishes the LSTM cell itself; h_t leaves the kernel in bf16, already the A
//     operand of the next step;
//   * the scheduled-sampling / straight-through feedback needs only the arg-max of the previous step's vocabulary row.  The fn_2
//     product's epilogue reduces every 64-column row piece it writes and folds it into one packed word per row with a 64-bit
//     atomic max (value in the high half, inverted column index in the low half: the largest logit, the FIRST index among equals, in
//     any arrival order); the next step's cst_dec_gates reads 8 bytes per row and gathers the embedding itself.  The softmax of the
//     soft decode (rnn.py:83) is off the recurrence's critical path: it runs once, after the loop, over all T steps.
//
// E = 128 (one row of the embedding table per 8 lanes), K = E + H_dec a multiple of 64, H_dec a multiple of 16; any batch size.
#include "cst_common.h"

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

__device__ __forceinline__ bf16_t dk_f2bf(float v) { __bf16 h = (__bf16)v; return __builtin_bit_cast(unsigned short, h); }
__device__ __forceinline__ unsigned dk_pack2(float a, float b) { return (unsigned)dk_f2bf(a) | ((unsigned)dk_f2bf(b) << 16); }
// byte offset of 16-byte slot `slot` (8 bf16 of k) of row `row` inside one K-tile image [rows][128 B] (the swizzle of gemm_bf16.hip)
__device__ __forceinline__ int dk_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }
// hardware exp / reciprocal forms (v_exp_f32, v_rcp_f32: ~1e-6 relative), as the whole-sequence encoder kernels use (lstm_seq.hip)
__device__ __forceinline__ float dk_sigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float dk_tanh(float x) { return 2.f * __frcp_rn(1.f + __expf(-2.f * x)) - 1.f; }
constexpr int DK_AMAX_GROUPS = 32;     // = CST_AMAX_GROUPS of gemm_bf16.hip (cst_argmax_groups())

// Whole-K panel [ROWS][K] bf16 -> LDS as K-tile images [kt][ROWS][128 B], by LDS-DMA: one wave instruction moves 8 rows x 128 B
// (lane l lands at row 8c + l / 8, physical slot l % 8 and therefore fetches logical slot (l % 8) ^ (row & 7)).  `rowptr(r)` = global
// address of tile row r.  Chunks kt0 * ROWS / 8 .. go round-robin over the NW waves; nothing waits here.
template <int ROWS, int NW, typename RowPtr>
__device__ __forceinline__ void dk_issue_panel(char* region, int kt0, int nkt, int wave, int lane, RowPtr rowptr) {
    constexpr int CPT = ROWS / 8;                           // chunks per K-tile
    const int lrow = lane >> 3, lps = lane & 7;
    for (int ch = kt0 * CPT + wave; ch < nkt * CPT; ch += NW) {
        const int kt = ch / CPT, rc = ch - kt * CPT;
        const int r = rc * 8 + lrow;
        const bf16_t* src = rowptr(r) + kt * 64 + ((lps ^ (r & 7)) << 3);
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(region + kt * ROWS * 128 + rc * 1024), 16, 0, 0);
    }
}

// acc[j] (+)= A[16 rows of wave row wm] . B[16 columns j of wave column wn]^T over the whole K; TN column tiles per wave
template <int BM, int BN, int TN, int NWN>
__device__ __forceinline__ void dk_mfma_tile(const char* As, const char* Bs, int kt, int wm, int wn, int lr, int lq, f32x4_t (&acc)[TN]) {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const u32x4_t a = *reinterpret_cast<const u32x4_t*>(As + kt * BM * 128 + dk_off(16 * wm + lr, kk * 4 + lq));
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const u32x4_t b = *reinterpret_cast<const u32x4_t*>(Bs + kt * BN * 128 + dk_off(wn * (BN / NWN) + 16 * j + lr, kk * 4 + lq));
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[j], 0, 0, 0);
        }
    }
}
// A wave has ~2 MFMAs per three 16-byte LDS reads here: the loop is LDS-latency-bound unless several K-tiles' reads are in flight.  With
// the tile count known at compile time (the reference's widths: K = 640 / 1024 / 512) the loop is fully unrolled and hipcc hoists the
// reads as far as the register file allows; other widths take the rolled loop (four tiles per trip).
template <int BM, int BN, int TN, int NWN, int KSTEP = 1>
__device__ __forceinline__ void dk_mfma_all(const char* As, const char* Bs, int nkt, int wm, int wn, int lane, f32x4_t (&acc)[TN], int kt0 = 0) {
    const int lr = lane & 15, lq = lane >> 4;
    if (nkt == 10) {
#pragma unroll
        for (int kt = 0; kt < 10; kt += KSTEP) dk_mfma_tile<BM, BN, TN, NWN>(As, Bs, kt0 + kt, wm, wn, lr, lq, acc);
    } else if (nkt == 16) {
#pragma unroll
        for (int kt = 0; kt < 16; kt += KSTEP) dk_mfma_tile<BM, BN, TN, NWN>(As, Bs, kt0 + kt, wm, wn, lr, lq, acc);
    } else if (nkt == 8) {
#pragma unroll
        for (int kt = 0; kt < 8; kt += KSTEP) dk_mfma_tile<BM, BN, TN, NWN>(As, Bs, kt0 + kt, wm, wn, lr, lq, acc);
    } else {
#pragma unroll 4
        for (int kt = kt0; kt < nkt; kt += KSTEP) dk_mfma_tile<BM, BN, TN, NWN>(As, Bs, kt, wm, wn, lr, lq, acc);
    }
}

// =============================================================================================
// cst_dec_gates
// =============================================================================================
struct DecGatesArgs {
    const bf16_t* A; long lda;                 // [B, K] bf16: [x_t | h_{t-1}]; the x part is only read when amax == null
    const bf16_t* W; long ldw;                 // [4 Hd, K] bf16, K contiguous: [W_ih | W_hh]
    const unsigned long long* amax;            // [DK_AMAX_GROUPS][B] packed arg-max words of the previous step's vocabulary rows, or null
    const int64_t* ids_t; long ldids;          // teacher tokens x[:, t-1] (column view) or null
    const int* coin;                           // device word: != 0 feeds the arg-max back, 0 the teacher token; null = always feed back
    const float* table; long ldtab; int V;     // token embedding [V, E] fp32
    CstDrop drop;                              // dropout of x_t over the (B, E) index space
    bf16_t* xb_out; long ldxb;                 // bf16 x_t kept for the weight-gradient product (may be A's own x columns)
    const float* bias;                         // [4 Hd] b_ih + b_hh
    const float* c_prev; long ldcp;
    float* gates; long ldg;                    // [B, 4 Hd] ACTIVATED gates i, f, g, o (what the backward pass reads)
    float* c_out; long ldc;
    float* h_out; long ldh;
    bf16_t* hb_next; long ldhb;                // bf16 h_t: the h columns of the next step's A (null at the last step)
    int B, E, Hd, K;
};

// 512 threads: 8 waves issue the DMA pieces (15 each at K = 640: a piece costs its wave 60-180 cycles to issue), then wave (wm, q) =
// (wave >> 2, wave & 3) multiplies rows 16 wm .. + 15 by the 16 columns of gate q.
__global__ __launch_bounds__(512) void dec_gates_kernel(DecGatesArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dk_smem[];
    const int nkt = a.K >> 6;
    char* As = dk_smem;                                     // [nkt][32][128 B]
    char* Bs = dk_smem + nkt * 32 * 128;                    // [nkt][64][128 B]
    const int m0 = blockIdx.x * 32, ny = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kt_x = a.amax ? (a.E >> 6) : 0;               // K-tiles of x_t filled by the gather below instead of DMA
    // weights: tile row 16 q + j = gate q of hidden unit 16 ny + j
    dk_issue_panel<64, 8>(Bs, 0, nkt, wave, lane, [&](int r) { return a.W + (long)((r >> 4) * a.Hd + 16 * ny + (r & 15)) * a.ldw; });
    // rows past the batch re-read the last valid row (their results are never stored)
    dk_issue_panel<32, 8>(As, kt_x, nkt, wave, lane, [&](int r) { return a.A + (long)min(m0 + r, a.B - 1) * a.lda; });
    if (a.amax) {
        // x_t = dropout(E[token]): 16 lanes per batch row, 8 elements each (E == 128); lane j of a row also owns group j of the row's
        // DK_AMAX_GROUPS packed arg-max words (j and j + 16)
        const int row = threadIdx.x >> 4, j = threadIdx.x & 15, c0 = j * 8;
        const bool live = m0 + row < a.B;
        const int b = min(m0 + row, a.B - 1);
        // the words that decide the token are requested together (one round trip, not three dependent ones; absent operands read a
        // valid dummy address instead of branching around their load: a branch would put one full wait per load)
        const int64_t* tp = a.ids_t ? a.ids_t + (long)b * a.ldids : reinterpret_cast<const int64_t*>(a.amax + b);
        const int* cp = a.coin ? a.coin : reinterpret_cast<const int*>(a.amax);
        unsigned long long pk = a.amax[(long)j * a.B + b], pk2 = a.amax[(long)(j + 16) * a.B + b];
        long tid = *tp;
        int coin_raw = *cp;
        asm volatile("" : "+v"(pk), "+v"(pk2), "+v"(tid), "+v"(coin_raw));      // all in registers before any of them is looked at
        pk = pk2 > pk ? pk2 : pk;
        static_assert(DK_AMAX_GROUPS == 32, "two group words per lane, 16 lanes per row");
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {                            // max over the 16 group words (16 consecutive lanes)
            const unsigned lo = __shfl_xor((unsigned)pk, o, 64), hi = __shfl_xor((unsigned)(pk >> 32), o, 64);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            pk = other > pk ? other : pk;
        }
        const int coin = a.coin ? coin_raw : 1;
        const long id = (a.ids_t && !coin) ? tid : (long)(0xFFFFFFFFu - (unsigned)(pk & 0xFFFFFFFFull));
        const bool ok = id >= 0 && id < a.V;
        const float4* trow = reinterpret_cast<const float4*>(a.table + (ok ? id : 0) * a.ldtab + c0);
        float4 v[2];
        v[0] = trow[0]; v[1] = trow[1];                               // unconditional (row 0 when the id is out of range), select after
        if (!ok) { v[0] = make_float4(0.f, 0.f, 0.f, 0.f); v[1] = v[0]; }
        if (a.drop.p > 0.f) {
            const uint32_t dseed = cst_drop_seed(a.drop);
            const uint32_t di = (uint32_t)((long)b * a.E + c0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                v[i].x *= cst_drop_mask(a.drop, dseed, di + 4 * i);     v[i].y *= cst_drop_mask(a.drop, dseed, di + 4 * i + 1);
                v[i].z *= cst_drop_mask(a.drop, dseed, di + 4 * i + 2); v[i].w *= cst_drop_mask(a.drop, dseed, di + 4 * i + 3);
            }
        }
        u32x4_t q;
        q.x = dk_pack2(v[0].x, v[0].y); q.y = dk_pack2(v[0].z, v[0].w); q.z = dk_pack2(v[1].x, v[1].y); q.w = dk_pack2(v[1].z, v[1].w);
        *reinterpret_cast<u32x4_t*>(As + (c0 >> 6) * 32 * 128 + dk_off(row, (c0 & 63) >> 3)) = q;
        if (ny == 0 && a.xb_out && live) *reinterpret_cast<u32x4_t*>(a.xb_out + (long)b * a.ldxb + c0) = q;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every DMA piece of this wave has landed ...
    __syncthreads();                                       // ... and every other wave's
    const int wm = wave >> 2, q = wave & 3;
    const int lr = lane & 15, lq = lane >> 4;
    f32x4_t acc[1] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}};
    dk_mfma_all<32, 64, 1, 4>(As, Bs, nkt, wm, q, lane, acc);
    __syncthreads();                                       // all fragment reads done: the A region becomes the exchange buffer
    // wave (wm, q) holds gate q of (row 16 wm + 4 lq + r, unit 16 ny + lr)
    float* ex = reinterpret_cast<float*>(dk_smem);         // [32 rows][16 units][4 gates]
#pragma unroll
    for (int r = 0; r < 4; ++r) ex[((16 * wm + 4 * lq + r) * 16 + lr) * 4 + q] = acc[0][r];
    __syncthreads();
    {
        const int row = threadIdx.x >> 4, ul = threadIdx.x & 15;          // one (row, unit) per thread
        const long b = m0 + row;
        if (b < a.B) {
            const int u = 16 * ny + ul, Hd = a.Hd;
            const float4 pre = *reinterpret_cast<const float4*>(ex + (row * 16 + ul) * 4);
            const float gi = dk_sigmoid(pre.x + a.bias[u]), gf = dk_sigmoid(pre.y + a.bias[Hd + u]);
            const float gg = dk_tanh(pre.z + a.bias[2 * Hd + u]), go = dk_sigmoid(pre.w + a.bias[3 * Hd + u]);
            const float c = gf * a.c_prev[b * a.ldcp + u] + gi * gg;
            const float h = go * dk_tanh(c);
            float* g = a.gates + b * a.ldg + u;
            g[0] = gi; g[Hd] = gf; g[2 * Hd] = gg; g[3 * Hd] = go;
            a.c_out[b * a.ldc + u] = c;
            a.h_out[b * a.ldh + u] = h;
            if (a.hb_next) a.hb_next[b * a.ldhb + u] = dk_f2bf(h);
        }
    }
}

extern "C" long cst_dec_gates_lds_bytes(int E, int Hd) { return (long)((E + Hd) / 64) * (32 + 64) * 128; }

extern "C" int cst_dec_gates(const void* A, long lda, const void* W, long ldw,
                             const void* amax_prev, const int64_t* ids_teacher, long ldids, const int* coin_dev,
                             const float* table, long ldtab, int V,
                             float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                             void* x_bf16_out, long ldxb, const float* bias, const float* c_prev, long ldcp,
                             float* gates, long ldg, float* c_out, long ldc, float* h_out, long ldh, void* h_bf16_next, long ldhb,
                             int B, int E, int Hd, void* stream) {
    CST_REQUIRE(A && W && bias && c_prev && gates && c_out && h_out, "cst_dec_gates: null pointer");
    const int K = E + Hd;
    CST_REQUIRE(B > 0 && Hd > 0 && Hd % 16 == 0 && E == 128 && K % 64 == 0,
                "cst_dec_gates: needs Hd %% 16 == 0, E == 128, (E + Hd) %% 64 == 0 (B=%d, E=%d, Hd=%d)", B, E, Hd);
    CST_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && ((((uintptr_t)A) | ((uintptr_t)W)) & 15) == 0,
                "cst_dec_gates: operands must be 16-byte aligned with leading dimensions >= K, multiples of 8");
    CST_REQUIRE(!amax_prev || (table && ldtab >= E && ldtab % 4 == 0 && (((uintptr_t)table) & 15) == 0 && V > 0 && (((uintptr_t)amax_prev) & 7) == 0),
                "cst_dec_gates: the gather needs a 16-byte aligned embedding table (ldtab %% 4 == 0) and 8-byte aligned arg-max words");
    CST_REQUIRE(!x_bf16_out || (ldxb % 8 == 0 && (((uintptr_t)x_bf16_out) & 15) == 0), "cst_dec_gates: x_bf16_out must be 16-byte aligned, ldxb %% 8 == 0");
    const long lds = cst_dec_gates_lds_bytes(E, Hd);
    CST_REQUIRE(lds <= 160 * 1024, "cst_dec_gates: K=%d needs %ld bytes of LDS (max 160 KiB)", K, lds);
    DecGatesArgs a;
    a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw;
    a.amax = (const unsigned long long*)amax_prev; a.ids_t = ids_teacher; a.ldids = ldids; a.coin = coin_dev;
    a.table = table; a.ldtab = ldtab; a.V = V;
    a.drop = cst_make_drop(amax_prev ? drop_p : 0.f, drop_seed, drop_stream, drop_seed_dev, (long)B * E);
    a.xb_out = (bf16_t*)x_bf16_out; a.ldxb = ldxb; a.bias = bias; a.c_prev = c_prev; a.ldcp = ldcp;
    a.gates = gates; a.ldg = ldg; a.c_out = c_out; a.ldc = ldc; a.h_out = h_out; a.ldh = ldh;
    a.hb_next = (bf16_t*)h_bf16_next; a.ldhb = ldhb; a.B = B; a.E = E; a.Hd = Hd; a.K = K;
    static CstPerDevice attr_done;
    if (cst_first_on_device(attr_done)) {
        (void)hipFuncSetAttribute((const void*)dec_gates_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(dec_gates_kernel, dim3((B + 31) / 32, Hd / 16), dim3(512), (size_t)lds, (hipStream_t)stream, a);
    CST_LAUNCH_CHECK("cst_dec_gates");
    return CST_OK;
}

// =============================================================================================
// cst_dec_attn: single-query dot attention of one decode step (rnn.py:46-50, :76) + dropout of the FFN input (rnn.py:79)
//
// One workgroup per batch row, thread t owns dimensions 2t, 2t + 1 of the 512: every encoder state of the row is requested straight
// into registers in ONE burst (L' independent 8-byte loads per thread, no LDS staging: the tile is read once and used twice -- for the
// scores and for the weighted sum), the L' scores are reduced by DPP + one LDS exchange, every thread finishes the softmax itself.
// One memory round trip and two barriers where cst_dot_attn_fwd stages the tile through LDS behind five.
// =============================================================================================
template <int LMAX>
__global__ __launch_bounds__(256) void dec_attn_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ mem,
                                                       float* __restrict__ out, long ldo, float* __restrict__ p, int L, float scale,
                                                       bf16_t* __restrict__ dropped_b, long lddropb, CstDrop drop) {
    constexpr int D = 512;
    __shared__ float red[4][LMAX];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float2 qv = *reinterpret_cast<const float2*>(q + (long)b * ldq + 2 * t);
    const float* mb = mem + (long)b * L * D + 2 * t;
    float2 m[LMAX];
#pragma unroll
    for (int j = 0; j < LMAX; ++j) m[j] = j < L ? *reinterpret_cast<const float2*>(mb + (long)j * D) : make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        if (j < L) {                                       // wave-uniform
            const float s = wave_sum(qv.x * m[j].x + qv.y * m[j].y);
            if (lane == 0) red[w][j] = s;
        }
    }
    __syncthreads();
    float sc[LMAX], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        sc[j] = j < L ? ((red[0][j] + red[1][j]) + (red[2][j] + red[3][j])) * scale : -INFINITY;
        mx = fmaxf(mx, sc[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) { sc[j] = j < L ? __expf(sc[j] - mx) : 0.f; sum += sc[j]; }
    const float inv = 1.f / sum;
    float2 o = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        const float pj = sc[j] * inv;
        o.x += pj * m[j].x; o.y += pj * m[j].y;
        if (t == j && j < L) p[(long)b * L + j] = pj;
    }
    *reinterpret_cast<float2*>(out + (long)b * ldo + 2 * t) = o;
    if (dropped_b) {                                       // dropout index space is the (B, 2D) matrix [q | out]
        float hq0 = qv.x, hq1 = qv.y, ho0 = o.x, ho1 = o.y;
        if (drop.p > 0.f) {
            const uint32_t dseed = cst_drop_seed(drop);
            const uint32_t i0 = (uint32_t)((long)b * 2 * D + 2 * t);
            hq0 *= cst_drop_mask(drop, dseed, i0); hq1 *= cst_drop_mask(drop, dseed, i0 + 1);
            ho0 *= cst_drop_mask(drop, dseed, i0 + D); ho1 *= cst_drop_mask(drop, dseed, i0 + D + 1);
        }
        unsigned* db = reinterpret_cast<unsigned*>(dropped_b + (long)b * lddropb);
        db[t] = dk_pack2(hq0, hq1);
        db[D / 2 + t] = dk_pack2(ho0, ho1);
    }
}

extern "C" int cst_dec_attn(const float* q, long ldq, const float* mem, float* out, long ldo, float* p, int B, int L, int D,
                            void* dropped_bf16, long lddropb,
                            float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream) {
    CST_REQUIRE(q && mem && out && p && B > 0, "cst_dec_attn: null pointer");
    CST_REQUIRE(D == 512 && L > 0 && L <= 64, "cst_dec_attn: needs D == 512 and L <= 64 (D=%d, L=%d)", D, L);
    CST_REQUIRE(ldq % 2 == 0 && ldo % 2 == 0 && lddropb % 2 == 0 && ((((uintptr_t)q) | ((uintptr_t)mem) | ((uintptr_t)out)) & 7) == 0 &&
                (((uintptr_t)dropped_bf16) & 3) == 0, "cst_dec_attn: rows must be 8-byte aligned");
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)B * 2 * D);
    const float scale = 1.0f / sqrtf((float)D);
    hipStream_t st = (hipStream_t)stream;
    if (L <= 24) hipLaunchKernelGGL(dec_attn_kernel<24>, dim3(B), dim3(256), 0, st, q, ldq, mem, out, ldo, p, L, scale, (bf16_t*)dropped_bf16, lddropb, dr);
    else if (L <= 40) hipLaunchKernelGGL(dec_attn_kernel<40>, dim3(B), dim3(256), 0, st, q, ldq, mem, out, ldo, p, L, scale, (bf16_t*)dropped_bf16, lddropb, dr);
    else hipLaunchKernelGGL(dec_attn_kernel<64>, dim3(B), dim3(256), 0, st, q, ldq, mem, out, ldo, p, L, scale, (bf16_t*)dropped_bf16, lddropb, dr);
    CST_LAUNCH_CHECK("cst_dec_attn");
    return CST_OK;
}

// =============================================================================================
// cst_gemm_bf16_skinny: C / Cb [M, N] = act(A[M, K] . B[N, K]^T + bias), whole K in LDS, 32 x 32 tiles, no K split
// =============================================================================================
struct SkinnyArgs {
    const bf16_t* A; long lda; const bf16_t* B; long ldb;
    float* C; long ldc; bf16_t* Cb; long ldcb;
    const float* bias; int act;               // 0 none, 1 relu, 2 LeakyReLU(0.1)
    CstDrop drop;                             // dropout over the (M, N) index space, applied last
    int M, N, K;
};

// 512 threads: 8 waves issue the DMA pieces; wave (kh, wm, wn) = (wave >> 2, (wave >> 1) & 1, wave & 1) multiplies the 16 x 16 output
// tile (wm, wn) over the K-tiles of parity kh, and the two halves are added through LDS in a fixed order (kh = 0 first).
__global__ __launch_bounds__(512) void gemm_bf16_skinny_kernel(SkinnyArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dk_smem[];
    const int nkt = a.K >> 6;
    char* As = dk_smem;
    char* Bs = dk_smem + nkt * 32 * 128;
    const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    dk_issue_panel<32, 8>(Bs, 0, nkt, wave, lane, [&](int r) { return a.B + (long)(n0 + r) * a.ldb; });
    dk_issue_panel<32, 8>(As, 0, nkt, wave, lane, [&](int r) { return a.A + (long)min(m0 + r, a.M - 1) * a.lda; });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int kh = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
    f32x4_t acc[1] = {(f32x4_t){0.f, 0.f, 0.f, 0.f}};
    if (nkt & 1) {                                         // odd tile count: the generic loop, wave half kh takes tiles kh, kh + 2, ...
        for (int kt = kh; kt < nkt; kt += 2) dk_mfma_tile<32, 32, 1, 2>(As, Bs, kt, wm, wn, lr, lq, acc);
    } else {
        dk_mfma_all<32, 32, 1, 2, 2>(As, Bs, nkt, wm, wn, lane, acc, kh);
    }
    __syncthreads();                                       // fragment reads done: reuse the A region
    float* ex = reinterpret_cast<float*>(dk_smem);         // [4 tiles][64 lanes][4]
    if (kh == 1) *reinterpret_cast<f32x4_t*>(ex + ((wave & 3) * 64 + lane) * 4) = acc[0];
    __syncthreads();
    if (kh == 1) return;
    const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(ex + ((wave & 3) * 64 + lane) * 4);
    const int n = n0 + 16 * wn + lr;
    const float bv = a.bias ? a.bias[n] : 0.f;
    const uint32_t dseed = a.drop.p > 0.f ? cst_drop_seed(a.drop) : 0u;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long m = m0 + 16 * wm + 4 * lq + r;
        if (m >= a.M) continue;
        float v = (acc[0][r] + hi[r]) + bv;
        if (a.act == 1) v = v > 0.f ? v : 0.f;
        else if (a.act == 2) v = v > 0.f ? v : 0.1f * v;
        if (a.drop.p > 0.f) v *= cst_drop_mask(a.drop, dseed, (uint32_t)(m * a.N + n));
        if (a.C) a.C[m * a.ldc + n] = v;
        if (a.Cb) a.Cb[m * a.ldcb + n] = dk_f2bf(v);
    }
}

extern "C" int cst_gemm_bf16_skinny(const void* A, long lda, const void* B, long ldb, float* C, long ldc, void* Cb, long ldcb,
                                    int M, int N, int K, const float* bias, int act,
                                    float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream) {
    CST_REQUIRE(A && B && (C || Cb), "cst_gemm_bf16_skinny: null operand");
    CST_REQUIRE(M > 0 && N > 0 && K > 0 && N % 32 == 0 && K % 64 == 0, "cst_gemm_bf16_skinny: N=%d must be a multiple of 32, K=%d of 64 (M=%d)", N, K, M);
    CST_REQUIRE((long)(K / 64) * 64 * 128 <= 160 * 1024, "cst_gemm_bf16_skinny: K=%d does not fit the LDS whole (max 1280)", K);
    CST_REQUIRE(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0,
                "cst_gemm_bf16_skinny: operands must be 16-byte aligned with leading dimensions >= K, multiples of 8");
    CST_REQUIRE((!C || ldc >= N) && (!Cb || ldcb >= N) && act >= 0 && act <= 2, "cst_gemm_bf16_skinny: bad output leading dimension or activation");
    SkinnyArgs a;
    a.A = (const bf16_t*)A; a.lda = lda; a.B = (const bf16_t*)B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.Cb = (bf16_t*)Cb; a.ldcb = ldcb;
    a.bias = bias; a.act = act; a.M = M; a.N = N; a.K = K;
    a.drop = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)M * N);
    static CstPerDevice attr_done;
    if (cst_first_on_device(attr_done)) {
        (void)hipFuncSetAttribute((const void*)gemm_bf16_skinny_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(gemm_bf16_skinny_kernel, dim3((M + 31) / 32, N / 32), dim3(512), (size_t)(K / 64) * 64 * 128, (hipStream_t)stream, a);
    CST_LAUNCH_CHECK("cst_gemm_bf16_skinny");
    return CST_OK;
}

// =============================================================================================
// cst_dec_fn2: the vocabulary projection of one decode step, logits[M, V] = r1[M, 512] . W2[V, 512]^T (rnn.py:80), plus the packed
// arg-max words of every row (see cst_gemm_bf16_argmax).
//
// The general NT kernel runs this 2.6 GFLOP product in 13-16 us: K = 512 is eight K-tiles fetched one tile ahead of 0.1 us of MFMA, so
// every tile costs an L2 round trip, and 316 small tiles each re-read their rows of r1.  Here r1 is STATIONARY: each of the four MFMA
// waves of a workgroup loads its 16 rows once, straight from global memory into MFMA A fragments (64 registers a lane, kept for the
// whole launch), and the workgroup walks a contiguous range of vocabulary columns in 32-column sub-tiles (32 KB of W2 each) through a
// ring of five LDS buffers -- all 160 KB, requested at once at the start, so the first five sub-tiles (the whole range at V = 10 000) cost
// ONE exposed round trip.  Roles are split as in a loader / consumer GEMM because gfx950 counts loads and stores in ONE in-order
// counter: waves 4-7 only issue LDS-DMA and count vmcnt (loads only), waves 0-3 only read fragments, issue MFMAs and store C (stores
// only, never waited for).  One barrier per sub-tile:
//   barrier(j) = "sub-tile j has landed" (the loaders waited for it) + "the MFMA waves are done with sub-tile j - 1" (they arrive after it)
//   loaders after barrier(j), j >= 1: DMA sub-tile j + 4 into buffer (j + 4) % 5 = the buffer sub-tile j - 1 left
// The arg-max of a row is kept per lane across the sub-tiles, reduced over the 16 lanes of a DPP row at the end and folded into the
// row's packed words with ONE atomic per row and workgroup.
// =============================================================================================
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ u32x4_t dk_lds_read128(unsigned addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
template <int CTRL>
__device__ __forceinline__ int dk_dpp_i(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int dk_row16_min_i(int v) {
    v = min(v, dk_dpp_i<0xB1>(v, v));
    v = min(v, dk_dpp_i<0x4E>(v, v));
    v = min(v, dk_dpp_i<0x141>(v, v));
    v = min(v, dk_dpp_i<0x140>(v, v));
    return v;
}
__device__ __forceinline__ unsigned long long dk_pack_max(float v, int idx) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx);
}

struct Fn2Args {
    const bf16_t* A; long lda;        // [M, 512] bf16
    const bf16_t* W; long ldw;        // [V, 512] bf16
    float* C; long ldc;               // [M, V] fp32
    unsigned long long* amax;         // [DK_AMAX_GROUPS][M] packed words (zeroed by the caller) or null
    int M, V, nsub;                   // nsub = 32-column sub-tiles per workgroup
    int vec4;                         // V, ldc multiples of 4 and C 16-byte aligned: 16-byte stores
};

constexpr int FN2_KT = 8;             // K = 512
constexpr int FN2_SUB = 32 * FN2_KT * 128;      // bytes of one W2 sub-tile image: 32 KB
constexpr int FN2_DMA = 8;            // DMA pieces per loader wave and sub-tile
constexpr int FN2_NB = 5;             // ring buffers: 5 x 32 KB = all 160 KB of LDS

__global__ __launch_bounds__(512) void dec_fn2_kernel(Fn2Args a) {
    extern __shared__ __attribute__((aligned(16))) char dk_smem[];
    auto buf = [&](int j) -> char* { return dk_smem + (j % FN2_NB) * FN2_SUB; };
    // The workgroups that share a slice of W2 (same column range, different row tile) should share an XCD's L2: the dispatcher deals
    // consecutive workgroup ids round-robin over the 8 XCDs, so id = 8 k + x puts (row tile k % mt, slice x + 8 (k / mt)) on XCD group x.
    // (gridDim.y is a multiple of 8; slices past the last sub-tile exit at once.)  Speed only -- any placement computes the same thing.
    const int mt = gridDim.x;
    const int id = blockIdx.x + mt * blockIdx.y, xg = id & 7, kq = id >> 3;
    const int mtile = kq % mt, slice = xg + 8 * (kq / mt);
    const int m0 = mtile * 64;
    const int sub0 = slice * a.nsub;                        // first sub-tile of this workgroup
    const int nsub_tot = (a.V + 31) >> 5;
    const int nsub = min(a.nsub, nsub_tot - sub0);
    if (nsub <= 0) return;                                  // the whole workgroup, before any barrier
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave >= 4) {
        // ------------------------------------------------------------------------------------------ loaders
        const int lw = wave - 4;
        auto issue_sub = [&](int j) {
            const int n0 = (sub0 + j) * 32;
            dk_issue_panel<32, 4>(buf(j), 0, FN2_KT, lw, lane, [&](int r) { return a.W + (long)min(n0 + r, a.V - 1) * a.ldw; });
        };
        for (int j = 0; j < min(nsub, FN2_NB); ++j) issue_sub(j);          // the whole ring at once: up to 160 KB in flight per CU
        for (int j = 0; j < nsub; ++j) {
            // sub-tile j must have landed; younger DMA of this wave: everything issued after it
            const int issued_hi = min(nsub - 1, max(FN2_NB - 1, j + FN2_NB - 2));
            const int younger = issued_hi - j;
            if (younger >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * FN2_DMA) : "memory");
            else if (younger == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * FN2_DMA) : "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * FN2_DMA) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FN2_DMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                   // barrier(j): sub-tile j is in LDS; the MFMA waves are done with sub-tile j - 1
            if (j >= 1 && j + FN2_NB - 1 < nsub) issue_sub(j + FN2_NB - 1);     // into the buffer sub-tile j - 1 left
        }
        return;
    }
    // ---------------------------------------------------------------------------------------------- MFMA waves
    const int lr = lane & 15, lq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)dk_smem;
    // this wave's 16 rows of r1 as MFMA A fragments, straight from global memory (read once per workgroup: no LDS round trip)
    u32x4_t af[2 * FN2_KT];
    {
        const bf16_t* arow = a.A + (long)min(m0 + 16 * wave + lr, a.M - 1) * a.lda + lq * 8;
#pragma unroll
        for (int ks = 0; ks < 2 * FN2_KT; ++ks) af[ks] = *reinterpret_cast<const u32x4_t*>(arow + ks * 32);
    }
    // Operands are swapped in the MFMAs below (W2 rows first): the 16 x 16 result comes out transposed, so a lane holds FOUR CONSECUTIVE
    // columns (4 lq + r) of ONE row (16 wave + lr) -- a 16-byte store per tile instead of four 4-byte ones, and one running maximum a lane.
    const long mrow = m0 + 16 * wave + lr;
    const bool row_ok = mrow < a.M;
    float bestv = -INFINITY;
    int besti = 0x7fffffff;
    for (int j = 0; j < nsub; ++j) {
        __builtin_amdgcn_s_barrier();                       // barrier(j): sub-tile j has landed
        const unsigned bb = lds_base + (unsigned)(buf(j) - dk_smem);
        f32x4_t acc0 = (f32x4_t){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
        u32x4_t bq[2][8];
        // group g = K steps 4g .. 4g + 3: 8 reads (4 K steps x 2 column tiles); the next group's reads fly under this group's MFMAs
        auto rd = [&](int ks, int ct) -> u32x4_t {          // fragment of K step ks (K-tile ks / 2, half ks % 2), column tile ct
            const unsigned base = bb + (unsigned)((ks >> 1) * 32 * 128);
            return dk_lds_read128(base + dk_off(16 * ct + lr, (ks & 1) * 4 + lq));
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) { bq[0][2 * q] = rd(q, 0); bq[0][2 * q + 1] = rd(q, 1); }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) { bq[(g + 1) & 1][2 * q] = rd(4 * (g + 1) + q, 0); bq[(g + 1) & 1][2 * q + 1] = rd(4 * (g + 1) + q, 1); }
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");          // this group's 8 reads are back, the next group's 8 may fly
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf16x8_t av = __builtin_bit_cast(bf16x8_t, af[4 * g + q]);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bq[g & 1][2 * q]), av, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bq[g & 1][2 * q + 1]), av, acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // C row 16 wave + lr, columns n0 + 4 lq + r (acc0) and n0 + 16 + 4 lq + r (acc1)
        const int n0 = (sub0 + j) * 32;
        const int c0 = n0 + 4 * lq, c1 = n0 + 16 + 4 * lq;
        if (row_ok) {
            if (a.vec4) {                                   // V and ldc multiples of 4, C 16-byte aligned: a group of four columns is whole or absent
                if (c0 < a.V) *reinterpret_cast<f32x4_t*>(a.C + mrow * a.ldc + c0) = acc0;
                if (c1 < a.V) *reinterpret_cast<f32x4_t*>(a.C + mrow * a.ldc + c1) = acc1;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (c0 + r < a.V) a.C[mrow * a.ldc + c0 + r] = acc0[r];
                    if (c1 + r < a.V) a.C[mrow * a.ldc + c1 + r] = acc1[r];
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c0 + r < a.V && acc0[r] > bestv) { bestv = acc0[r]; besti = c0 + r; }
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c1 + r < a.V && acc1[r] > bestv) { bestv = acc1[r]; besti = c1 + r; }
    }
    if (a.amax) {
        // a lane saw its columns in ascending order, so its strict > kept its FIRST maximal column; the four lanes of a row (lq = 0 .. 3) then
        // agree on (largest value, smallest column) and lane lq = 0 folds it into the row's group word
        const int grp = slice & (DK_AMAX_GROUPS - 1);
#pragma unroll
        for (int sh = 16; sh <= 32; sh <<= 1) {
            const float ov = __shfl_xor(bestv, sh);
            const int oi = __shfl_xor(besti, sh);
            if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
        }
        if (lq == 0 && row_ok) atomicMax(a.amax + (long)grp * a.M + mrow, dk_pack_max(bestv, besti));
    }
}

extern "C" int cst_dec_fn2(const void* A, long lda, const void* W, long ldw, float* C, long ldc, int M, int V, int K,
                           void* amax_packed, void* stream) {
    CST_REQUIRE(A && W && C && M > 0 && V > 0, "cst_dec_fn2: bad arguments");
    CST_REQUIRE(K == 64 * FN2_KT, "cst_dec_fn2: K=%d (this kernel is built for K = %d)", K, 64 * FN2_KT);
    CST_REQUIRE(lda >= K && ldw >= K && lda % 8 == 0 && ldw % 8 == 0 && ((((uintptr_t)A) | ((uintptr_t)W)) & 15) == 0 && ldc >= V,
                "cst_dec_fn2: operands must be 16-byte aligned with leading dimensions >= K, multiples of 8; ldc >= V");
    CST_REQUIRE(!amax_packed || (((uintptr_t)amax_packed) & 7) == 0, "cst_dec_fn2: arg-max words must be 8-byte aligned");
    Fn2Args a;
    a.A = (const bf16_t*)A; a.lda = lda; a.W = (const bf16_t*)W; a.ldw = ldw; a.C = C; a.ldc = ldc;
    a.amax = (unsigned long long*)amax_packed; a.M = M; a.V = V;
    a.vec4 = (V % 4 == 0 && ldc % 4 == 0 && (((uintptr_t)C) & 15) == 0) ? 1 : 0;
    const int mt = (M + 63) / 64, nsub_tot = (V + 31) / 32;
    int slices = 256 / mt; if (slices < 1) slices = 1; if (slices > nsub_tot) slices = nsub_tot;
    a.nsub = (nsub_tot + slices - 1) / slices;
    slices = ((nsub_tot + a.nsub - 1) / a.nsub + 7) / 8 * 8;        // a multiple of 8 (XCD grouping in the kernel); surplus slices exit at once
    static CstPerDevice attr_done;
    if (cst_first_on_device(attr_done)) {
        (void)hipFuncSetAttribute((const void*)dec_fn2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    hipLaunchKernelGGL(dec_fn2_kernel, dim3(mt, slices), dim3(512), (size_t)FN2_NB * FN2_SUB, (hipStream_t)stream, a);
    CST_LAUNCH_CHECK("cst_dec_fn2");
    return CST_OK;
}

// =============================================================================================
// packed arg-max words -> token ids
// =============================================================================================
__global__ void unpack_argmax_kernel(const unsigned long long* __restrict__ packed, int64_t* __restrict__ ids, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    packed += (long)blockIdx.y * DK_AMAX_GROUPS * n;      // step blockIdx.y
    unsigned long long best = 0ull;
#pragma unroll
    for (int g = 0; g < DK_AMAX_GROUPS; ++g) {
        const unsigned long long w = packed[(long)g * n + i];
        best = w > best ? w : best;
    }
    ids[(long)blockIdx.y * n + i] = (int64_t)(0xFFFFFFFFu - (unsigned)(best & 0xFFFFFFFFull));
}

// packed: `steps` blocks of [cst_argmax_groups()][n] words (one block per cst_gemm_bf16_argmax product), ids [steps][n]
extern "C" int cst_unpack_argmax(const void* packed, int64_t* ids, long n, int steps, void* stream) {
    CST_REQUIRE(packed && ids && n > 0 && steps > 0 && steps < 65536, "cst_unpack_argmax: bad arguments");
    hipLaunchKernelGGL(unpack_argmax_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)steps), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)packed, ids, n);
    CST_LAUNCH_CHECK("cst_unpack_argmax");
    return CST_OK;
}

// =============================================================================================
// cst_dec_attn_cell_bwd: one step of the soft decode's backward behind the fn_1 dgrad -- the single-query attention backward
// (rnn.py:46-50) and the LSTM cell backward (rnn.py:75) of one batch row in one workgroup.
//
//   g = d[h_s | a_s] (the dropped FFN-input gradient);  dp_j = g_a . mem_j;  ds_j = p_j (dp_j - sum_i p_i dp_i) / sqrt(D);
//   d h_s = g_h + sum_j ds_j mem_j + dh2 (the recurrent gradient from step s + 1)  ->  dgates_s, dc_{s-1}.
//
// Thread t owns dimensions 2t, 2t + 1 as in dec_attn_kernel: the row's encoder states go straight into registers, are used for the
// L' dot products and for the weighted sum, and the cell backward of the thread's two hidden units follows in the same registers.
// d memory is NOT touched here: the step stores its L' values ds_j, and cst_dec_attn_dmem adds every step's p_j g_a + ds_j h_s in
// one launch after the loop -- the per-step read-modify-write of the (L', D) gradient tile (2/3 of the old kernel's traffic) is gone.
// =============================================================================================
struct AttnCellBwdArgs {
    const float* g; long ldg; const float* mem; const float* p; float* ds_out;
    const float* gates; long ldgt; const float* c_prev; long ldcp; const float* c_new; long ldcn;
    const float* dh2; long lddh2; const float* dc; long lddc;
    float* dgates; long lddg; float* dc_prev; long lddcp; bf16_t* dgb; long lddgb;
    int L; float scale;
};

template <int LMAX>
__global__ __launch_bounds__(256) void dec_attn_cell_bwd_kernel(AttnCellBwdArgs a) {
    constexpr int D = 512;
    __shared__ float red[4][LMAX];
    __shared__ float ps[LMAX];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6, L = a.L;
    const float* gb = a.g + (long)b * a.ldg;
    const float2 gh = *reinterpret_cast<const float2*>(gb + 2 * t);
    const float2 ga = *reinterpret_cast<const float2*>(gb + D + 2 * t);
    const float pv = a.p[(long)b * L + min(t, L - 1)];
    // the cell's operands travel with the memory tile: one round trip for everything the row needs
    const float* gt = a.gates + (long)b * a.ldgt + 2 * t;
    const float2 gi = *reinterpret_cast<const float2*>(gt), gf = *reinterpret_cast<const float2*>(gt + D);
    const float2 gg = *reinterpret_cast<const float2*>(gt + 2 * D), go = *reinterpret_cast<const float2*>(gt + 3 * D);
    const float2 cp = *reinterpret_cast<const float2*>(a.c_prev + (long)b * a.ldcp + 2 * t);
    const float2 cn = *reinterpret_cast<const float2*>(a.c_new + (long)b * a.ldcn + 2 * t);
    const float2 d2 = a.dh2 ? *reinterpret_cast<const float2*>(a.dh2 + (long)b * a.lddh2 + 2 * t) : make_float2(0.f, 0.f);
    const float2 dcv = a.dc ? *reinterpret_cast<const float2*>(a.dc + (long)b * a.lddc + 2 * t) : make_float2(0.f, 0.f);
    const float* mb = a.mem + (long)b * L * D + 2 * t;
    float2 m[LMAX];
#pragma unroll
    for (int j = 0; j < LMAX; ++j) m[j] = j < L ? *reinterpret_cast<const float2*>(mb + (long)j * D) : make_float2(0.f, 0.f);
    if (t < L) ps[t] = pv;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        if (j < L) {                                       // wave-uniform
            const float s = wave_sum(ga.x * m[j].x + ga.y * m[j].y);
            if (lane == 0) red[w][j] = s;
        }
    }
    __syncthreads();
    float dsj[LMAX], delta = 0.f;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        dsj[j] = j < L ? (red[0][j] + red[1][j]) + (red[2][j] + red[3][j]) : 0.f;
        delta += j < L ? ps[j] * dsj[j] : 0.f;
    }
    float2 dq = make_float2(0.f, 0.f);
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        const float d = j < L ? ps[j] * (dsj[j] - delta) * a.scale : 0.f;
        dq.x += d * m[j].x; dq.y += d * m[j].y;
        if (t == j && j < L) a.ds_out[(long)b * L + j] = d;
    }
    const float dh[2] = {gh.x + dq.x + d2.x, gh.y + dq.y + d2.y};
    const float iv[2] = {gi.x, gi.y}, fv[2] = {gf.x, gf.y}, gv[2] = {gg.x, gg.y}, ov[2] = {go.x, go.y};
    const float cpv[2] = {cp.x, cp.y}, cnv[2] = {cn.x, cn.y}, dci[2] = {dcv.x, dcv.y};
    float g0[2], g1[2], g2[2], g3[2], dcp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {                          // as lstm_cell_bwd_kernel (pointwise.hip)
        const float tc = tanhf(cnv[u]);
        const float dct = dci[u] + dh[u] * ov[u] * (1.f - tc * tc);
        g0[u] = dct * gv[u] * iv[u] * (1.f - iv[u]);
        g1[u] = dct * cpv[u] * fv[u] * (1.f - fv[u]);
        g2[u] = dct * iv[u] * (1.f - gv[u] * gv[u]);
        g3[u] = dh[u] * tc * ov[u] * (1.f - ov[u]);
        dcp[u] = dct * fv[u];
    }
    float* dg = a.dgates + (long)b * a.lddg + 2 * t;
    *reinterpret_cast<float2*>(dg) = make_float2(g0[0], g0[1]);
    *reinterpret_cast<float2*>(dg + D) = make_float2(g1[0], g1[1]);
    *reinterpret_cast<float2*>(dg + 2 * D) = make_float2(g2[0], g2[1]);
    *reinterpret_cast<float2*>(dg + 3 * D) = make_float2(g3[0], g3[1]);
    if (a.dgb) {
        unsigned* q = reinterpret_cast<unsigned*>(a.dgb + (long)b * a.lddgb);
        q[t] = dk_pack2(g0[0], g0[1]); q[D / 2 + t] = dk_pack2(g1[0], g1[1]);
        q[D + t] = dk_pack2(g2[0], g2[1]); q[3 * D / 2 + t] = dk_pack2(g3[0], g3[1]);
    }
    *reinterpret_cast<float2*>(a.dc_prev + (long)b * a.lddcp + 2 * t) = make_float2(dcp[0], dcp[1]);
}

extern "C" int cst_dec_attn_cell_bwd(const float* g, long ldg, const float* mem, const float* p, float* ds_out, int B, int L, int D,
                                     const float* gates, long ldgt, const float* c_prev, long ldcp, const float* c_new, long ldcn,
                                     const float* dh2, long lddh2, const float* dc, long lddc,
                                     float* dgates, long lddg, float* dc_prev, long lddcp, void* dgates_bf16, long lddgb, void* stream) {
    CST_REQUIRE(g && mem && p && ds_out && gates && c_prev && c_new && dgates && dc_prev && B > 0, "cst_dec_attn_cell_bwd: null pointer");
    CST_REQUIRE(D == 512 && L > 0 && L <= 64, "cst_dec_attn_cell_bwd: needs D == 512 and L <= 64 (D=%d, L=%d)", D, L);
    CST_REQUIRE(ldg >= 2 * D && ldgt >= 4 * D && lddg >= 4 * D && (!dgates_bf16 || lddgb >= 4 * D), "cst_dec_attn_cell_bwd: leading dimension too small");
    const long lds_or = ldg | ldgt | ldcp | ldcn | lddg | lddcp | (dh2 ? lddh2 : 0) | (dc ? lddc : 0) | (dgates_bf16 ? lddgb : 0);
    const uintptr_t ptr_or = (uintptr_t)g | (uintptr_t)mem | (uintptr_t)gates | (uintptr_t)c_prev | (uintptr_t)c_new | (uintptr_t)dh2 | (uintptr_t)dc |
                             (uintptr_t)dgates | (uintptr_t)dc_prev;
    CST_REQUIRE(lds_or % 2 == 0 && (ptr_or & 7) == 0 && (((uintptr_t)dgates_bf16) & 3) == 0, "cst_dec_attn_cell_bwd: rows must be 8-byte aligned");
    AttnCellBwdArgs a{g, ldg, mem, p, ds_out, gates, ldgt, c_prev, ldcp, c_new, ldcn, dh2, lddh2, dc, lddc, dgates, lddg, dc_prev, lddcp,
                      (bf16_t*)dgates_bf16, lddgb, L, 1.0f / sqrtf((float)D)};
    hipStream_t st = (hipStream_t)stream;
    if (L <= 24) hipLaunchKernelGGL(dec_attn_cell_bwd_kernel<24>, dim3(B), dim3(256), 0, st, a);
    else if (L <= 40) hipLaunchKernelGGL(dec_attn_cell_bwd_kernel<40>, dim3(B), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(dec_attn_cell_bwd_kernel<64>, dim3(B), dim3(256), 0, st, a);
    CST_LAUNCH_CHECK("cst_dec_attn_cell_bwd");
    return CST_OK;
}

// d memory of all T steps of the soft decode in one launch: dmem[b, j, :] += sum_s p[s, b, j] g_a[b, s, :] + ds[s, b, j] h[b, s, :]
// (the two outer products cst_dot_attn_bwd adds per step).  One workgroup per batch row, thread t owns dimensions 2t, 2t + 1.
template <int LMAX>
__global__ __launch_bounds__(256) void dec_attn_dmem_kernel(const float* __restrict__ ga, long ldga, long ga_step, const float* __restrict__ h, long ldh, long h_step,
                                                            const float* __restrict__ p, const float* __restrict__ ds, float* __restrict__ dmem,
                                                            int B, int T, int L) {
    constexpr int D = 512;
    extern __shared__ float pd[];                          // [T][2][L]
    const int b = blockIdx.x, t = threadIdx.x;
    for (int e = t; e < T * L; e += 256) {
        const int s = e / L, j = e % L;
        pd[(s * 2) * L + j] = p[((long)s * B + b) * L + j];
        pd[(s * 2 + 1) * L + j] = ds[((long)s * B + b) * L + j];
    }
    __syncthreads();
    float2 acc[LMAX];
#pragma unroll
    for (int j = 0; j < LMAX; ++j) acc[j] = make_float2(0.f, 0.f);
    const float* gr = ga + (long)b * ldga + 2 * t;
    const float* hr = h + (long)b * ldh + 2 * t;
    float2 gv = *reinterpret_cast<const float2*>(gr), hv = *reinterpret_cast<const float2*>(hr);
    for (int s = 0; s < T; ++s) {
        const int sn = min(s + 1, T - 1);
        const float2 gn = *reinterpret_cast<const float2*>(gr + sn * ga_step), hn = *reinterpret_cast<const float2*>(hr + sn * h_step);
        const float* pr = pd + (s * 2) * L;
#pragma unroll
        for (int j = 0; j < LMAX; ++j) {
            if (j < L) {
                const float pj = pr[j], dj = pr[L + j];
                acc[j].x += pj * gv.x + dj * hv.x; acc[j].y += pj * gv.y + dj * hv.y;
            }
        }
        gv = gn; hv = hn;
    }
    float* o = dmem + (long)b * L * D + 2 * t;
#pragma unroll
    for (int j = 0; j < LMAX; ++j) {
        if (j < L) {
            float2 v = *reinterpret_cast<float2*>(o + (long)j * D);
            v.x += acc[j].x; v.y += acc[j].y;
            *reinterpret_cast<float2*>(o + (long)j * D) = v;
        }
    }
}

// ga: g_a of step s, row b at ga[b * ldga + s * ga_step + 0..D); h likewise; p, ds: [T][B][L]; dmem [B][L][D] accumulated into
extern "C" int cst_dec_attn_dmem(const float* ga, long ldga, long ga_step, const float* h, long ldh, long h_step,
                                 const float* p, const float* ds, float* dmem, int B, int T, int L, int D, void* stream) {
    CST_REQUIRE(ga && h && p && ds && dmem && B > 0 && T > 0, "cst_dec_attn_dmem: null pointer");
    CST_REQUIRE(D == 512 && L > 0 && L <= 64 && (size_t)T * 2 * L * sizeof(float) <= 64 * 1024, "cst_dec_attn_dmem: needs D == 512, L <= 64 (D=%d, L=%d, T=%d)", D, L, T);
    CST_REQUIRE((ldga | ga_step | ldh | h_step) % 2 == 0 && ((((uintptr_t)ga) | ((uintptr_t)h) | ((uintptr_t)dmem)) & 7) == 0, "cst_dec_attn_dmem: rows must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)T * 2 * L * sizeof(float);
    if (L <= 24) hipLaunchKernelGGL(dec_attn_dmem_kernel<24>, dim3(B), dim3(256), lds, st, ga, ldga, ga_step, h, ldh, h_step, p, ds, dmem, B, T, L);
    else if (L <= 40) hipLaunchKernelGGL(dec_attn_dmem_kernel<40>, dim3(B), dim3(256), lds, st, ga, ldga, ga_step, h, ldh, h_step, p, ds, dmem, B, T, L);
    else hipLaunchKernelGGL(dec_attn_dmem_kernel<64>, dim3(B), dim3(256), lds, st, ga, ldga, ga_step, h, ldh, h_step, p, ds, dmem, B, T, L);
    CST_LAUNCH_CHECK("cst_dec_attn_dmem");
    return CST_OK;
}

// =============================================================================================
// cst_dec_dxe: the straight-through gradient of the soft decode (rnn.py:84-85), one step: C[M, V] += dropout(g)[M, 128] . E[V, 128]^T
//
// 0.66 GFLOP against a read-modify-write of the step's (M, V) gradient block: an HBM-bound accumulate with a product inside.
// The 16 batch rows of a wave are the MFMA's B operand (dropout applied and rounded to bf16 on the way into registers, kept for the
// whole launch; the workgroups of slice 0 also write the dropped fp32 rows -- the operand of the embedding scatter after the loop), each
// wave walks the 32-column chunks of the workgroup's vocabulary slice with the table rows as the A operand (LDS, see the kernel).  MFMA row i of
// tile f is vocabulary column v0 + 8 (i / 4) + 4 f + i % 4, so a lane's eight accumulators are eight CONSECUTIVE columns of one batch
// row: the tile of C is read into the accumulators and written back as whole 128-byte lines per four lanes.
// =============================================================================================
struct DxeArgs {
    const float* g; long ldg; float* gx; long ldgx; const bf16_t* E; long lde; float* C; long ldc;
    int M, V, cps, rb, ns; CstDrop drop;
};

// 64 batch rows per workgroup: wave w owns rows 16w .. 16w + 15 and all four waves walk the SAME 32-column chunks of the workgroup's
// vocabulary slice.  The slice's table rows (cps x 32 rows x 256 B) are requested whole by LDS-DMA at the start (two K-tile images,
// the swizzle of dk_off) next to the first two C tiles, and the dropout masks are computed while they travel.
__global__ __launch_bounds__(256) void dec_dxe_kernel(DxeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char dk_smem[];
    const int lane = threadIdx.x & 63, lr = lane & 15, lq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // workgroup id -> (row block, vocabulary slice), XCD-aware: consecutive ids go round-robin over the 8 XCDs, so XCD x takes the x-th
    // eighth of the slice-major order and its L2 fetches one eighth of the table instead of all of it
    const int per = (a.rb * a.ns + 7) >> 3;
    const int logical = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    if (logical >= a.rb * a.ns) return;
    const int slice = logical / a.rb, rblk = logical - slice * a.rb;
    const int b0 = rblk * 64 + 16 * wave;
    const bool row_ok = b0 + lr < a.M;
    const long brow = min(b0 + lr, a.M - 1);
    const int nchunks = (a.V + 31) >> 5;
    const int c_lo = slice * a.cps, c_hi = min(c_lo + a.cps, nchunks);
    const int rows = (c_hi - c_lo) * 32, v_lo = c_lo * 32;
    {   // table rows -> LDS: one wave instruction = 8 rows x 128 B of one K-tile
        const int lrow = lane >> 3, lps = lane & 7;
        const int npieces = 2 * (rows >> 3);
        for (int pc = wave; pc < npieces; pc += 4) {
            const int kt = pc & 1, rc = pc >> 1;
            const int r = rc * 8 + lrow;
            const bf16_t* src = a.E + (long)min(v_lo + r, a.V - 1) * a.lde + kt * 64 + ((lps ^ (r & 7)) << 3);
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dk_smem + kt * rows * 128 + rc * 1024), 16, 0, 0);
        }
    }
    float* crow = a.C + brow * a.ldc;
    f32x4_t acc[2][2];                                     // [chunk parity][f]
    auto load_c = [&](int c, f32x4_t (&dst)[2]) {
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int v = c * 32 + 8 * lq + 4 * f;
            dst[f] = (row_ok && v < a.V) ? *reinterpret_cast<const f32x4_t*>(crow + v) : (f32x4_t){0.f, 0.f, 0.f, 0.f};
        }
    };
    load_c(c_lo, acc[0]);
    if (c_lo + 1 < c_hi) load_c(c_lo + 1, acc[1]);
    const uint32_t dseed = a.drop.p > 0.f ? cst_drop_seed(a.drop) : 0u;
    const bool write_gx = a.gx && slice == 0 && row_ok;
    bf16x8_t gb[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int k0 = 32 * kk + 8 * lq;
        const float4* src = reinterpret_cast<const float4*>(a.g + brow * a.ldg + k0);
        float4 x = src[0], y = src[1];
        if (a.drop.p > 0.f) {
            const uint32_t i0 = (uint32_t)(brow * 128 + k0);
            x.x *= cst_drop_mask(a.drop, dseed, i0); x.y *= cst_drop_mask(a.drop, dseed, i0 + 1);
            x.z *= cst_drop_mask(a.drop, dseed, i0 + 2); x.w *= cst_drop_mask(a.drop, dseed, i0 + 3);
            y.x *= cst_drop_mask(a.drop, dseed, i0 + 4); y.y *= cst_drop_mask(a.drop, dseed, i0 + 5);
            y.z *= cst_drop_mask(a.drop, dseed, i0 + 6); y.w *= cst_drop_mask(a.drop, dseed, i0 + 7);
        }
        if (write_gx) {
            float4* o = reinterpret_cast<float4*>(a.gx + brow * a.ldgx + k0);
            o[0] = x; o[1] = y;
        }
        u32x4_t pk;
        pk.x = dk_pack2(x.x, x.y); pk.y = dk_pack2(x.z, x.w); pk.z = dk_pack2(y.x, y.y); pk.w = dk_pack2(y.z, y.w);
        gb[kk] = __builtin_bit_cast(bf16x8_t, pk);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the table rows (and the first C tiles) have landed
    __syncthreads();
    auto do_chunk = [&](int c, f32x4_t (&cur)[2]) {
        f32x4_t t[2] = {cur[0], cur[1]};
        if (c + 2 < c_hi) load_c(c + 2, cur);              // two C tiles ahead: the wait for it leaves this chunk's and the next one's stores in flight
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int r = (c - c_lo) * 32 + 8 * (lr >> 2) + 4 * f + (lr & 3);    // MFMA row lr of tile f
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const u32x4_t e = *reinterpret_cast<const u32x4_t*>(dk_smem + (kk >> 1) * rows * 128 + dk_off(r, (kk & 1) * 4 + lq));
                t[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, e), gb[kk], t[f], 0, 0, 0);
            }
        }
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            const int v = c * 32 + 8 * lq + 4 * f;
            if (row_ok && v < a.V) *reinterpret_cast<f32x4_t*>(crow + v) = t[f];
        }
    };
    for (int c = c_lo; c < c_hi; c += 2) {
        do_chunk(c, acc[0]);
        if (c + 1 < c_hi) do_chunk(c + 1, acc[1]);
    }
}

// g [M, 128] fp32 (row stride ldg); gx_out (optional) receives dropout(g); E_bf16 [V, lde >= 128]; C [M, V] (row stride ldc) accumulated into.
// Dropout over the (M, 128) index space, as cst_dropout on the same matrix.  V and ldc multiples of 4.
extern "C" int cst_dec_dxe(const float* g, long ldg, float* gx_out, long ldgx, const void* E_bf16, long lde, float* C, long ldc,
                           int M, int V, int K, float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream) {
    CST_REQUIRE(g && E_bf16 && C && M > 0 && V > 0, "cst_dec_dxe: null pointer");
    CST_REQUIRE(K == 128 && V % 4 == 0 && ldc % 4 == 0 && ldc >= V && ldg % 4 == 0 && ldg >= K && lde % 8 == 0 && lde >= K && (!gx_out || (ldgx % 4 == 0 && ldgx >= K)),
                "cst_dec_dxe: needs K == 128, V and the leading dimensions multiples of 4 (K=%d, V=%d)", K, V);
    CST_REQUIRE(((((uintptr_t)g) | ((uintptr_t)gx_out) | ((uintptr_t)E_bf16) | ((uintptr_t)C)) & 15) == 0, "cst_dec_dxe: operands must be 16-byte aligned");
    DxeArgs a;
    a.g = g; a.ldg = ldg; a.gx = gx_out; a.ldgx = ldgx; a.E = (const bf16_t*)E_bf16; a.lde = lde; a.C = C; a.ldc = ldc; a.M = M; a.V = V;
    a.drop = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)M * K);
    const int rb = (M + 63) / 64, nchunks = (V + 31) / 32;
    // ~one workgroup per CU: the per-workgroup prologue (g rows, masks) is a third of the launch (8.7 us at 256 / 512 workgroups, 9.8 at 768)
    int ns = (256 + rb - 1) / rb;
    if (ns > (nchunks + 1) / 2) ns = (nchunks + 1) / 2;     // two chunks per workgroup or more
    ns = ns < 1 ? 1 : ns;
    a.cps = (nchunks + ns - 1) / ns;
    if (a.cps > 16) a.cps = 16;                             // 16 chunks = 128 KB of table rows in LDS
    ns = (nchunks + a.cps - 1) / a.cps;
    a.rb = rb; a.ns = ns;
    static CstPerDevice attr_done;
    if (cst_first_on_device(attr_done)) {
        (void)hipFuncSetAttribute((const void*)dec_dxe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    }
    hipLaunchKernelGGL(dec_dxe_kernel, dim3((rb * ns + 7) / 8 * 8), dim3(256), (size_t)a.cps * 32 * 256, (hipStream_t)stream, a);
    CST_LAUNCH_CHECK("cst_dec_dxe");
    return CST_OK;
}
