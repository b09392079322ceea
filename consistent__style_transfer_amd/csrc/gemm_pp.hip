// bf16 NT GEMM for the encoder-layer products, round 4: big tiles on eight waves that run in two groups half a phase apart
// ("ping-pong"), one persistent workgroup per CU, the K-tile stream continuing across output tiles.
//
//   C[M,N] = epilogue( alpha * A[M,K] . B[N,K]^T ),  operands bf16, K contiguous and a multiple of 64 (as cst_gemm_bf16).
//
// Why another kernel: the LDS-DMA tile kernels of gemm_bf16.hip (64x128 / 128x128 tiles, two or three workgroups
// per CU, one barrier per K-tile) sit at 19-29 % of the bf16 MFMA peak on the 9216- and 4608-row products of the Matcher and the MLM
// (mlm.py:20-22, match.py:18-20): their small tiles move 2-3x the L2->LDS bytes of a 256-wide tile (the fill runs at the L2's own
// rate), and whole-K-tile ring stages leave at most one K-tile of loads in flight.  This kernel is the structure
// cdna_hip_programming.md section 5 describes for deep-pipelined schedules, rebuilt for these shapes:
//
//  * tile (32 TM) x (64 TN), TM in 4..8 row tiles and TN in 2..4 column tiles of 16 per wave, eight waves as 2 (M) x 4 (N): the tile
//    shape is CHOSEN PER PRODUCT so that the tile count fills whole rounds of the 256 CUs (ragged edge tiles allowed) -- 9216 x 2304
//    runs as 42 x 12 = 504 tiles of 224 x 192 (two rounds at 96 %) where 256 x 256 tiles would be 324 (two rounds at 63 %);
//  * a K-tile (64 deep) is four PHASES, one per quadrant of the wave tile: (q0,q0) (q0,q1) (q1,q1) (q1,q0).  A phase = fragment reads
//    + LDS-DMA issue ("mem part"), barrier, MFMAs of the quadrant, barrier.  Waves 0-3 (row half 0) and 4-7 (row half 1) run one
//    barrier apart: while one group's MFMAs hold the matrix pipes the other group reads and issues.  Two waves per SIMD, one of each group;
//  * the two LDS stages are recycled REGION BY REGION: the A rows of a quadrant, and the B rows of a quadrant, are refilled with the
//    K-tile two ahead as soon as their last fragment read has retired (fragments live in registers), so 5-6 phases of loads are in
//    flight (8-10 KiB a wave) behind counted s_waitcnt vmcnt -- never a drain inside the stream;
//  * LDS image: [rows][128 B] with the 16-byte-slot XOR swizzle of gemm_bf16.hip (applied to each lane's SOURCE address: an LDS-DMA
//    instruction writes 1 KiB linearly); B rows are stored in MFMA order of a COLUMN-PERMUTED tile, so that after the swapped-operand
//    MFMA (C^T per 16 x 16 tile) a lane holds 4 TN CONSECUTIVE columns of one row: the epilogue stores 16-byte vectors straight
//    from the accumulators (no staging through LDS, so it does not disturb the ring: the next tile's K-tiles are already in flight).
//
// Barrier bookkeeping (b_k = k-th barrier instance; phase s = 4 u + p of K-tile u): group 0 executes B1(s) = b_2s, B2(s) = b_2s+1;
// group 1 starts with one extra barrier, so its B1(s) = b_2s+1, B2(s) = b_2s+2.  Reads of phase s are complete (lgkmcnt(0) after B1)
// before the issuing group's B2(s).  Regions and their refills (K-tile u+2 into the stage of K-tile u):
//     A(group g, q0): read by group g in P0;  refilled by group g in the mem part of P1   (after its B2(P0))
//     B(all, q0)    : read by both in P0, kept in registers through P3; refilled in P2       (after b_8u+2)
//     A(group g, q1): read in P2;             refilled by group g in P3
//     B(all, q1)    : read in P1;             refilled in P0 of K-tile u+1                   (after b_8u+4)
// and the two counted waits per K-tile (W_A in P3, W_B in P0, both in front of B1) are placed so that every wave has waited for a
// region's loads before the barrier that precedes its first read ("read a staged buffer one phase after the wait that retires it").
#include "bgemm.h"
#include <hip/hip_ext.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

namespace {

struct PpGeom {
    int tilesM, tilesN, ntiles, nkt, gn;
    int pm;         // column order inside a wave's 16 TN columns (see pp_col): 0 = natural (fp32 output), 1 = pairs of tiles interleaved (bf16 output)
    unsigned long long* stamps;   // bench build: 16 words per recording wave (s_memtime at section boundaries), or null
    int abl;        // timing ablations (CST_PP_ABL, tools/gemm_pp_bench.py abl; WRONG results): 1 no LDS-DMA, 2 no MFMA, 4 no fragment reads,
                    // 8 no second barrier of a phase, 16 no vmcnt waits, 32 no epilogue
};

__device__ __forceinline__ unsigned pp_bf16_2(float a, float b) {
    __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}
__device__ __forceinline__ float pp_bf2f(unsigned h) { return __uint_as_float(h << 16); }

// The lane id, recomputed where it is needed (tile changes, the epilogue): a value derived from threadIdx.x at kernel entry would sit in
// a VGPR -- with everything hipcc derives from it -- through the whole K loop, next to 128 accumulator and 64 fragment registers.
__device__ __forceinline__ int pp_lane() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

template <int OFF>
__device__ __forceinline__ u32x4_t pp_ds_read(unsigned addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// fragment reads of NT tiles (16 rows = 2 KiB apart), both 32-wide k halves: f[t][kk]
// (T0 = first tile of the quadrant: its 2 KiB steps ride in the instruction's offset field, so both quadrants share one address pair)
template <int T, int NT, int T0>
struct PpReadTiles {
    static __device__ __forceinline__ void run(u32x4_t (*f)[2], unsigned ad0, unsigned ad1) {
        f[T][0] = pp_ds_read<(T0 + T) * 2048>(ad0);
        f[T][1] = pp_ds_read<(T0 + T) * 2048>(ad1);
        PpReadTiles<T + 1, NT, T0>::run(f, ad0, ad1);
    }
};
template <int NT, int T0>
struct PpReadTiles<NT, NT, T0> {
    static __device__ __forceinline__ void run(u32x4_t (*)[2], unsigned, unsigned) {}
};

// wait for every LDS read of the wave; the fragments are named as read-write operands so that no use of them (and no copy the
// register allocator might want) can be scheduled above the wait (cdna_hip_programming.md 5.7, form ii)
template <int N>
__device__ __forceinline__ void pp_lgkm_fence(u32x4_t (*f)[2]) {
    if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1])::"memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1])::"memory");
    else if constexpr (N == 3)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1])::"memory");
    else if constexpr (N == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f[0][0]), "+v"(f[0][1]), "+v"(f[1][0]), "+v"(f[1][1]), "+v"(f[2][0]), "+v"(f[2][1]), "+v"(f[3][0]), "+v"(f[3][1])::"memory");
    else static_assert(N == 0, "fragment sets of up to four tiles");
}

template <int N>
__device__ __forceinline__ void pp_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// counted wait with `extra` store instructions of the last epilogue still inside the window (extra is one of the few sums of S0..S3)
template <int N, int S0, int S1, int S2, int S3>
__device__ __forceinline__ void pp_wait(int extra) {
    constexpr int c0 = N + S0 < 63 ? N + S0 : 63, c1 = N + S1 < 63 ? N + S1 : 63, c2 = N + S2 < 63 ? N + S2 : 63, c3 = N + S3 < 63 ? N + S3 : 63;
    if (extra == 0) pp_vmcnt<N>();
    else if (extra == S0) pp_vmcnt<c0>();
    else if (extra == S1) pp_vmcnt<c1>();
    else if (extra == S2) pp_vmcnt<c2>();
    else if (extra == S3) pp_vmcnt<c3>();
    else pp_vmcnt<0>();                                           // unknown count (a ragged tile's epilogue): drain
}

// Column held in register r of tile j by lane quarter lq = pp_col(j, lq) + r, inside a wave's 16 TN columns.  After the swapped-operand MFMA
// a lane holds 4 consecutive columns (r) of ONE row per 16 x 16 tile; WHICH columns a tile's MFMA row c = 4 lq + r stands for is free (it is
// only the order in which B rows are laid into LDS), and it decides how an epilogue store instruction lands in memory -- the store path
// processes a wave instruction line piece by line piece, so every instruction should write whole 64-byte pieces:
//   PM 0 (fp32 output): natural order, column 16 j + 4 lq + r: a float4 store of tile j writes 16 rows x 64 contiguous bytes;
//   PM 1 (bf16 output): tiles (2 t, 2 t + 1) interleaved, column 32 t + 8 lq + 4 (j & 1) + r: ONE 16-byte store per pair and lane,
//        16 rows x 64 contiguous bytes again (a last unpaired tile keeps the natural order: 8-byte stores).
template <int TN>
__device__ __forceinline__ int pp_col(int pm, int j, int lq) {
    if (pm == 1 && (j | 1) < TN) return 32 * (j >> 1) + 8 * lq + 4 * (j & 1);
    return 16 * j + 4 * lq;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// epilogue of one wave tile, straight from the accumulators: acc[i][j][r] = C[mw + 16 i + lr][nw + pp_col(j, lq) + r].  WHOLE: the wave
// tile lies inside the matrix (no per-lane tests: the number of store instructions is exact, which the counted waits after it rely on).
// ---------------------------------------------------------------------------------------------------------------------------------
template <int TM, int TN, bool WHOLE>
__device__ __forceinline__ void pp_epilogue(const BGemmArgs& g, f32x4_t (&acc)[TM][TN], int mw, int nw, int lr, int lq, int pm, uint32_t dseed) {
    const int act = g.act;
    const float alpha = g.alpha;
    const bool has_bias = g.bias != nullptr, has_add = g.addend != nullptr, has_aux = act >= 3, has_old = g.C && g.accumulate;
    const bool has_drop = g.drop.p > 0.f;
    // v > 0 ? v * ps : v * ns   (nz: the negative side is an exact +0)
    const float ps = act == 3 ? g.gate_scale : 1.f, ns = (act == 2 || act == 4) ? 0.1f : (act == 0 ? 1.f : 0.f);
    const bool nz = act == 1 || act == 3;
    int nj[TN];
    float b4[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        nj[j] = nw + pp_col<TN>(pm, j, lq);
        b4[j][0] = b4[j][1] = b4[j][2] = b4[j][3] = 0.f;
        if (has_bias && (WHOLE || nj[j] < g.N)) {
            const float4 t = *reinterpret_cast<const float4*>(g.bias + nj[j]);
            b4[j][0] = t.x; b4[j][1] = t.y; b4[j][2] = t.z; b4[j][3] = t.w;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int m = mw + 16 * i + lr;
        const bool okm = WHOLE || m < g.M;
        float4 ad[TN], old[TN];
        uint2 ax[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) { ad[j] = make_float4(0.f, 0.f, 0.f, 0.f); old[j] = ad[j]; ax[j] = make_uint2(0u, 0u); }
        if (has_add) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (okm && (WHOLE || nj[j] < g.N)) ad[j] = *reinterpret_cast<const float4*>(g.addend + (long)m * g.ldadd + nj[j]);
        }
        if (has_aux) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (okm && (WHOLE || nj[j] < g.N)) ax[j] = *reinterpret_cast<const uint2*>(g.aux + (long)m * g.ldaux + nj[j]);
        }
        if (has_old) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
                if (okm && (WHOLE || nj[j] < g.N)) old[j] = *reinterpret_cast<const float4*>(g.C + (long)m * g.ldc + nj[j]);
        }
        unsigned pk[TN][2];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float w = alpha * acc[i][j][e] + b4[j][e];
                if (has_add) w += (&ad[j].x)[e];
                float gv = w;
                if (has_aux) gv = pp_bf2f(e < 2 ? (ax[j].x >> (16 * e)) & 0xffffu : (ax[j].y >> (16 * (e - 2))) & 0xffffu);
                o[e] = gv > 0.f ? w * ps : (nz ? 0.f : w * ns);
            }
            if (has_drop) {
                const uint32_t di = (uint32_t)((long)m * g.N + nj[j]);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] *= cst_drop_mask(g.drop, dseed, di + e);
            }
            if (g.C) {
                if (has_old) { o[0] += old[j].x; o[1] += old[j].y; o[2] += old[j].z; o[3] += old[j].w; }
                if (okm && (WHOLE || nj[j] < g.N)) *reinterpret_cast<float4*>(g.C + (long)m * g.ldc + nj[j]) = make_float4(o[0], o[1], o[2], o[3]);
            }
            pk[j][0] = pp_bf16_2(o[0], o[1]);
            pk[j][1] = pp_bf16_2(o[2], o[3]);
        }
        if (g.Cb) {
            bf16_t* cb = g.Cb + (long)m * g.ldcb;
            if (pm == 1) {
#pragma unroll
                for (int j = 0; j + 1 < TN; j += 2)               // a pair = 8 consecutive columns (N % 8 == 0 is required for PM 1: whole or nothing)
                    if (okm && (WHOLE || nj[j] < g.N)) *reinterpret_cast<uint4*>(cb + nj[j]) = make_uint4(pk[j][0], pk[j][1], pk[j + 1][0], pk[j + 1][1]);
                if constexpr (TN % 2 == 1)
                    if (okm && (WHOLE || nj[TN - 1] < g.N)) *reinterpret_cast<uint2*>(cb + nj[TN - 1]) = make_uint2(pk[TN - 1][0], pk[TN - 1][1]);
            } else {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    if (okm && (WHOLE || nj[j] < g.N)) *reinterpret_cast<uint2*>(cb + nj[j]) = make_uint2(pk[j][0], pk[j][1]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same epilogue through a 4 KiB transposing buffer per wave (LDS beyond the two ring stages, so the ring is not disturbed).  Why: the
// store path takes a wave instruction piece by piece, and what it is given matters -- written from the accumulator layout an
// instruction is 16 rows x 64 bytes (9 GB/s a CU measured: 8.7 us for a 160 x 256 bf16 tile, a third of the launch); after the
// transposition a wave instruction is 4 rows x 256 bytes (fp32) / 4 rows x 128 bytes (bf16) of whole lines, every per-element operand
// (addend, gate, old C) is read as whole lines too, and a lane's bias is four registers.  Per 16-row tile of the wave tile: TN
// ds_write_b128 of the raw sums (row lr, 16-byte slot 4 j + lq, slots XOR-ed with the row so that neither side has bank conflicts),
// TN ds_read_b128 (piece p = 64 it + lane of the 16 x 4 TN pieces, row-major), the epilogue arithmetic on 4 consecutive columns, the
// stores.  LDS instructions of one wave execute in order, so the buffer needs no barrier (it is this wave's alone); they are inline asm,
// invisible to hipcc's "LDS-DMA in flight -> vmcnt(0) before any LDS access" rule.
// ---------------------------------------------------------------------------------------------------------------------------------
// LOWREG (128-register builds): the per-element operands are fetched piece by piece instead of a 16-row tile at a time
template <int TM, int TN, bool WHOLE, bool LOWREG>
__device__ __forceinline__ void pp_epilogue_lds(const BGemmArgs& g, f32x4_t (&acc)[TM][TN], unsigned cbuf, int mw, int nw, int lane, int pm, uint32_t dseed) {
    constexpr int NP = 4 * TN;                                   // 16-byte pieces (4 columns) per row
    constexpr int XM = TN == 4 ? 15 : (TN == 2 ? 7 : 0);         // slot swizzle (16 / 8 slots a row); TN == 3: padded rows instead
    constexpr int RS = TN == 3 ? NP * 16 + 16 : NP * 16;         // row stride in bytes
    static_assert(16 * RS <= 4096, "a 16-row tile fits the wave's buffer");
    const int lr = lane & 15, lq = lane >> 4;
    const int act = g.act;
    const float alpha = g.alpha;
    const bool has_add = g.addend != nullptr, has_aux = act >= 3, has_old = g.C && g.accumulate, has_drop = g.drop.p > 0.f;
    const float ps = act == 3 ? g.gate_scale : 1.f, ns = (act == 2 || act == 4) ? 0.1f : (act == 0 ? 1.f : 0.f);
    const bool nz = act == 1 || act == 3;
    unsigned wad[TN], rad[TN];
    int prow[TN], pn[TN];
    float b4[TN][4];
#pragma unroll
    for (int j = 0; j < TN; ++j) wad[j] = cbuf + lr * RS + ((((pp_col<TN>(pm, j, lq)) >> 2) ^ (lr & XM)) << 4);   // slot = column / 4 under either column order
#pragma unroll
    for (int it = 0; it < TN; ++it) {
        const int p = 64 * it + lane, row = p / NP, slot = p - row * NP;
        prow[it] = row;
        pn[it] = nw + 4 * slot;
        rad[it] = cbuf + row * RS + ((slot ^ (row & XM)) << 4);
        b4[it][0] = b4[it][1] = b4[it][2] = b4[it][3] = 0.f;
        if (g.bias && (WHOLE || pn[it] < g.N)) {
            const float4 t = *reinterpret_cast<const float4*>(g.bias + pn[it]);
            b4[it][0] = t.x; b4[it][1] = t.y; b4[it][2] = t.z; b4[it][3] = t.w;
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("ds_write_b128 %0, %1" ::"v"(wad[j]), "v"(acc[i][j]) : "memory");
        f32x4_t v[TN];
#pragma unroll
        for (int it = 0; it < TN; ++it) asm volatile("ds_read_b128 %0, %1" : "=v"(v[it]) : "v"(rad[it]) : "memory");
        // the per-element operands of this 16-row tile, requested while the LDS round trip is under way
        float4 ad[TN], old[TN];
        uint2 ax[TN];
        bool ok[TN];
#pragma unroll
        for (int it = 0; it < TN; ++it) {
            const int m = mw + 16 * i + prow[it];
            ok[it] = WHOLE || (m < g.M && pn[it] < g.N);
            ad[it] = make_float4(0.f, 0.f, 0.f, 0.f); old[it] = ad[it]; ax[it] = make_uint2(0u, 0u);
            if constexpr (!LOWREG) {
                if (has_add && ok[it]) ad[it] = *reinterpret_cast<const float4*>(g.addend + (long)m * g.ldadd + pn[it]);
                if (has_aux && ok[it]) ax[it] = *reinterpret_cast<const uint2*>(g.aux + (long)m * g.ldaux + pn[it]);
                if (has_old && ok[it]) old[it] = *reinterpret_cast<const float4*>(g.C + (long)m * g.ldc + pn[it]);
            }
        }
        if constexpr (TN == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
        else if constexpr (TN == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2])::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1])::"memory");
#pragma unroll
        for (int it = 0; it < TN; ++it) {
            const int m = mw + 16 * i + prow[it];
            if constexpr (LOWREG) {
                if (has_add && ok[it]) ad[it] = *reinterpret_cast<const float4*>(g.addend + (long)m * g.ldadd + pn[it]);
                if (has_aux && ok[it]) ax[it] = *reinterpret_cast<const uint2*>(g.aux + (long)m * g.ldaux + pn[it]);
                if (has_old && ok[it]) old[it] = *reinterpret_cast<const float4*>(g.C + (long)m * g.ldc + pn[it]);
            }
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float w = alpha * v[it][e] + b4[it][e];
                if (has_add) w += (&ad[it].x)[e];
                float gv = w;
                if (has_aux) gv = pp_bf2f(e < 2 ? (ax[it].x >> (16 * e)) & 0xffffu : (ax[it].y >> (16 * (e - 2))) & 0xffffu);
                o[e] = gv > 0.f ? w * ps : (nz ? 0.f : w * ns);
            }
            if (has_drop) {
                const uint32_t di = (uint32_t)((long)m * g.N + pn[it]);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] *= cst_drop_mask(g.drop, dseed, di + e);
            }
            if (g.C) {
                if (has_old) { o[0] += old[it].x; o[1] += old[it].y; o[2] += old[it].z; o[3] += old[it].w; }
                if (ok[it]) *reinterpret_cast<float4*>(g.C + (long)m * g.ldc + pn[it]) = make_float4(o[0], o[1], o[2], o[3]);
            }
            if (g.Cb && ok[it]) *reinterpret_cast<uint2*>(g.Cb + (long)m * g.ldcb + pn[it]) = make_uint2(pp_bf16_2(o[0], o[1]), pp_bf16_2(o[2], o[3]));
        }
    }
}

template <int TM, int TN>
struct PpCfg {
    static constexpr int TM0 = (TM + 1) / 2, TM1 = TM - TM0, TN0 = (TN + 1) / 2, TN1 = TN - TN0;
    static constexpr int BM = 32 * TM, BN = 64 * TN;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
    static constexpr int N1 = (2 * TM0 + 3) / 4, N3 = (2 * TM1 + 3) / 4;      // A pieces (8 rows) a wave issues for its own row half: q0, q1
    static constexpr int W_A = N3 + TN1 + N1 + TN0 + N3;                      // loads that may stay in flight at the P3 wait
    static constexpr int W_B = N1 + TN0 + N3 + TN1;                           // ... at the P0 wait
    // store instructions of a whole wave tile: fp32; bf16 in natural order (8-byte pieces); bf16 in pair order (16-byte pieces)
    static constexpr int SC = TM * TN, SB0 = TM * TN, SB1 = TM * (TN / 2 + TN % 2);
    static constexpr int LDS = 2 * STAGE;                                    // the ring
    static constexpr int CBUF = 8 * 4096;                                    // + a 4 KiB transposing buffer per wave (pp_epilogue_lds)
    static_assert(TM >= 2 && TM <= 8 && TN >= 2 && TN <= 4, "wave tile out of range");
    static_assert(LDS <= 160 * 1024, "two stages must fit the LDS");
};

enum { PP_FIRST = 0, PP_STEADY = 1, PP_PRELAST = 2, PP_LAST = 3 };
template <int V> struct PpMode { static constexpr int value = V; };

// OCC = workgroups per CU the kernel is built for: 1 (256 registers a lane, up to 160 KB of LDS) or 2 (128 registers, up to 80 KB: two
// workgroups out of phase, one's epilogue and prologue under the other's K loop)
// DBG (bench build only): 0 = the product kernel, 1 = + cycle stamps, 2 = + run-time ablation switches (which slow the whole kernel down)
template <int TM, int TN, int OCC = 1, int DBG = 0>
__global__ __launch_bounds__(512, 2 * OCC) void cst_gemm_bf16_pp_kernel(BGemmArgs g, PpGeom q) {
    const int abl = DBG == 2 ? q.abl : 0;
    using Cf = PpCfg<TM, TN>;
    constexpr int TM0 = Cf::TM0, TM1 = Cf::TM1, TN0 = Cf::TN0, TN1 = Cf::TN1, BM = Cf::BM, BN = Cf::BN;
    constexpr int A_BYTES = Cf::A_BYTES, STAGE = Cf::STAGE, N1 = Cf::N1, N3 = Cf::N3;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;                     // row half (= ping-pong group), column quarter
    const bool g1 = wr == 1;
    const uint32_t dseed = g.drop.p > 0.f ? cst_drop_seed(g.drop) : 0u;     // (a load: before the first LDS-DMA is in flight)
    constexpr bool LDS_EPI = OCC == 1;                            // one workgroup per CU: room for the transposing buffers (PpCfg::CBUF)
    static_assert(!LDS_EPI || Cf::LDS + Cf::CBUF <= 160 * 1024, "ring + transposing buffers must fit the LDS");
    const int pm = LDS_EPI ? 0 : q.pm;
    // DBG: cycle stamps of waves 0 and 4 of the first two logical workgroups at the section boundaries (start, after the prologue, after
    // every tile's K loop, after every epilogue), in a buffer nothing else reads (cdna_hip_programming.md section 7, In-kernel stamps)
    unsigned long long stamp[DBG ? 12 : 1];
    int nstamp = 0;
    auto mark = [&]() {
        if constexpr (DBG) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (nstamp < 12) stamp[nstamp] = t;
            ++nstamp;
        }
    };
    unsigned long long rt0 = 0;
    if constexpr (DBG) { asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt0)::"memory"); }
    mark();

    // ---- this workgroup's tiles: logical id (contiguous per XCD), then id, id + G, id + 2 G ...
    const int G = gridDim.x;
    int lid;
    {
        const int qd = G >> 3, rm = G & 7, xcd = blockIdx.x & 7;
        lid = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + (blockIdx.x >> 3);
    }
    const int my_tiles = lid < q.ntiles ? (q.ntiles - lid + G - 1) / G : 0;
    if (my_tiles == 0) return;
    const int nkt = q.nkt;                                        // >= 3 (bgemm_pp_try)
    auto tile_mn = [&](int r, int& m0, int& n0) {                // r-th tile of this workgroup -> its origin
        const int t = r * G + lid;
        const int grp = t / (q.gn * q.tilesM);
        const int gw = min(q.gn, q.tilesN - grp * q.gn);
        const int local = t - grp * q.gn * q.tilesM;
        const int tm = local / gw, tn = grp * q.gn + local - tm * gw;
        m0 = tm * BM; n0 = tn * BN;
    };

    // ---- LDS-DMA pieces of this wave (8 rows x 128 B each).  Lane l lands at (row + l / 8, physical slot l % 8) and fetches logical
    // slot (l % 8) ^ (row & 7).  A: LDS row = tile row.  B: LDS row p of column block cb = p / (16 TN) holds the tile column that MFMA row
    // p % 16 of tile (p / 16) % TN stands for (pp_col).
    const int wq = wave & 3, wcb = wave >> 1, wh = wave & 1;
    int rA0[N1], rA1[N3 > 0 ? N3 : 1], rB0[TN0], rB1[TN1 > 0 ? TN1 : 1];     // first LDS row of each piece (wave-uniform)
#pragma unroll
    for (int c = 0; c < N1; ++c) rA0[c] = wr * 16 * TM + 8 * min(wq * N1 + c, 2 * TM0 - 1);
#pragma unroll
    for (int c = 0; c < N3; ++c) rA1[c] = wr * 16 * TM + 16 * TM0 + 8 * min(wq * N3 + c, 2 * TM1 - 1);
#pragma unroll
    for (int c = 0; c < TN0; ++c) rB0[c] = wcb * 16 * TN + 8 * (wh * TN0 + c);
#pragma unroll
    for (int c = 0; c < TN1; ++c) rB1[c] = wcb * 16 * TN + 16 * TN0 + 8 * (wh * TN1 + c);
    // per-lane BYTE offsets from g.A / g.B (32 bits: operands of up to 4 GiB), so that a piece costs one VGPR and the load takes the
    // uniform base (+ the K-tile's offset) from scalar registers
    auto a_src = [&](int m0, int row) {
        const int lane = pp_lane(), lrow = lane >> 3, lps = lane & 7;
        row += lrow;
        return (unsigned)(((long)min(m0 + row, g.M - 1) * g.lda + ((lps ^ (row & 7)) << 3)) * 2);
    };
    auto b_src = [&](int n0, int p) {
        const int lane = pp_lane(), lrow = lane >> 3, lps = lane & 7;
        p += lrow;
        const int cb = p / (16 * TN), pp = p - cb * 16 * TN;
        const int nl = cb * 16 * TN + pp_col<TN>(pm, pp >> 4, (pp & 15) >> 2) + (pp & 3);
        return (unsigned)(((long)min(n0 + nl, g.N - 1) * g.ldb + ((lps ^ (p & 7)) << 3)) * 2);
    };
    // four load groups: D0 = B(q1), D1 = A(own, q0), D2 = B(q0), D3 = A(own, q1)
    unsigned pD0[TN1 > 0 ? TN1 : 1], pD1[N1], pD2[TN0], pD3[N3 > 0 ? N3 : 1];
    auto setD0 = [&](int r) { int m0, n0; tile_mn(r, m0, n0); _Pragma("unroll") for (int c = 0; c < TN1; ++c) pD0[c] = b_src(n0, rB1[c]); };
    auto setD1 = [&](int r) { int m0, n0; tile_mn(r, m0, n0); _Pragma("unroll") for (int c = 0; c < N1; ++c) pD1[c] = a_src(m0, rA0[c]); };
    auto setD2 = [&](int r) { int m0, n0; tile_mn(r, m0, n0); _Pragma("unroll") for (int c = 0; c < TN0; ++c) pD2[c] = b_src(n0, rB0[c]); };
    auto setD3 = [&](int r) { int m0, n0; tile_mn(r, m0, n0); _Pragma("unroll") for (int c = 0; c < N3; ++c) pD3[c] = a_src(m0, rA1[c]); };
    const char* const Ab = reinterpret_cast<const char*>(g.A);
    const char* const Bb = reinterpret_cast<const char*>(g.B);
    // kb = byte offset of the K-tile inside its rows (128 B per K-tile), so = byte offset of the destination stage
    auto issueD0 = [&](int kb, int so) {
        if (abl & 1) return;
        const char* base = Bb + kb;
        asm volatile("" : "+s"(base));                          // one scalar base: the load takes it as saddr, the lane's 32-bit offset as vaddr
        asm volatile("" : "+s"(base));                          // one scalar base: the load takes it as saddr, the lane's 32-bit offset as vaddr
#pragma unroll
        for (int c = 0; c < TN1; ++c) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + pD0[c]), (lds_ptr_t)(smem + so + A_BYTES + rB1[c] * 128), 16, 0, 0);
    };
    auto issueD1 = [&](int kb, int so) {
        if (abl & 1) return;
        const char* base = Ab + kb;
        asm volatile("" : "+s"(base));                          // one scalar base: the load takes it as saddr, the lane's 32-bit offset as vaddr
        asm volatile("" : "+s"(base));                          // one scalar base: the load takes it as saddr, the lane's 32-bit offset as vaddr
#pragma unroll
        for (int c = 0; c < N1; ++c) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + pD1[c]), (lds_ptr_t)(smem + so + rA0[c] * 128), 16, 0, 0);
    };
    auto issueD2 = [&](int kb, int so) {
        if (abl & 1) return;
        const char* base = Bb + kb;
        asm volatile("" : "+s"(base));                          // one scalar base: the load takes it as saddr, the lane's 32-bit offset as vaddr
        asm volatile("" : "+s"(base));                          // one scalar base: the load takes it as saddr, the lane's 32-bit offset as vaddr
#pragma unroll
        for (int c = 0; c < TN0; ++c) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + pD2[c]), (lds_ptr_t)(smem + so + A_BYTES + rB0[c] * 128), 16, 0, 0);
    };
    auto issueD3 = [&](int kb, int so) {
        if (abl & 1) return;
        const char* base = Ab + kb;
        asm volatile("" : "+s"(base));                          // one scalar base: the load takes it as saddr, the lane's 32-bit offset as vaddr
        asm volatile("" : "+s"(base));                          // one scalar base: the load takes it as saddr, the lane's 32-bit offset as vaddr
#pragma unroll
        for (int c = 0; c < N3; ++c) __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + pD3[c]), (lds_ptr_t)(smem + so + rA1[c] * 128), 16, 0, 0);
    };

    // ---- fragment read addresses of the CURRENT stage (tile t adds 2 KiB in the offset field; kk = 1 flips slot bit 2 = byte 64); they
    // move to the other stage by +/- STAGE after every K-tile
    unsigned fa_k0, fa_k1, fb_k0, fb_k1;
    {
        const int lane = pp_lane(), lr = lane & 15, lq = lane >> 4;
        const unsigned fa0 = (wr * 16 * TM + lr) * 128 + ((lq ^ (lr & 7)) << 4);
        const unsigned fb0 = A_BYTES + (wc * 16 * TN + lr) * 128 + ((lq ^ (lr & 7)) << 4);
        fa_k0 = lds0 + fa0; fa_k1 = lds0 + (fa0 ^ 64u);
        fb_k0 = lds0 + fb0; fb_k1 = lds0 + (fb0 ^ 64u);
    }
    int sdelta = STAGE;                                           // fragment addresses: to the other stage
    int so_cur = 0, so_oth = STAGE;                               // LDS byte offsets of the current K-tile's stage and of the other one

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    u32x4_t af[TM0][2], bf0[TN0][2], bf1[TN1 > 0 ? TN1 : 1][2];

    // ---- prologue: K-tile 0 whole, K-tile 1 but for its B(q1) (which P0 of K-tile 0 issues): the state every K-tile finds
    setD0(0); setD1(0); setD2(0); setD3(0);
    issueD1(0, 0); issueD2(0, 0); issueD3(0, 0); issueD0(0, 0);
    issueD1(128, STAGE); issueD2(128, STAGE); issueD3(128, STAGE);
    pp_vmcnt<Cf::W_A>();                                          // A(q0), B(q0) of K-tile 0 have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (g1) __builtin_amdgcn_s_barrier();                         // the stagger: group 1 runs one barrier behind
    mark();

    // ---- one K-tile = four phases.  MODE (compile time): FIRST = K-tile 0 of an output tile (the waits count the previous tile's epilogue
    // stores, `pend`), STEADY = every load of this K-tile belongs to the same output tile, PRELAST / LAST = K-tiles nkt - 2 / nkt - 1: their
    // loads cross into the next output tile of this workgroup (`nxt` >= 0: its index; < 0: the stream ends).
    auto ktile = [&](auto mode_c, int kt, int nxt, int pend) {
        constexpr int MODE = decltype(mode_c)::value;
        const bool has_next = nxt >= 0;
        // ---------------- P0: quadrant (q0, q0); load B(q1) of the next K-tile into the other stage
        if (!(abl & 4)) PpReadTiles<0, TN0, 0>::run(bf0, fb_k0, fb_k1);
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 4)) PpReadTiles<0, TM0, 0>::run(af, fa_k0, fa_k1);
        if constexpr (MODE == PP_LAST) {
            if (has_next) { setD0(nxt); issueD0(0, so_oth); }
        } else {
            issueD0((kt + 1) * 128, so_oth);
        }
        if (!(abl & 16)) {
            if constexpr (MODE == PP_FIRST) pp_wait<Cf::W_B, Cf::SC, Cf::SB1, Cf::SB0, Cf::SC + Cf::SB0>(pend);
            else if constexpr (MODE == PP_LAST) { if (has_next) pp_vmcnt<Cf::W_B>(); else pp_vmcnt<0>(); }
            else pp_vmcnt<Cf::W_B>();
        }
        __builtin_amdgcn_s_barrier();
        pp_lgkm_fence<TM0>(af);
        pp_lgkm_fence<TN0>(bf0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (!(abl & 2))
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < TM0; ++i)
#pragma unroll
                for (int j = 0; j < TN0; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf0[j][kk]), __builtin_bit_cast(bf16x8_t, af[i][kk]), acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 8)) __builtin_amdgcn_s_barrier();
        // ---------------- P1: quadrant (q0, q1); load A(own, q0) of the K-tile two ahead into this stage
        if constexpr (TN1 > 0) { if (!(abl & 4)) PpReadTiles<0, TN1, TN0>::run(bf1, fb_k0, fb_k1); }
        if constexpr (MODE == PP_PRELAST) {
            if (has_next) { setD1(nxt); issueD1(0, so_cur); }
        } else if constexpr (MODE == PP_LAST) {
            if (has_next) issueD1(128, so_cur);
        } else {
            issueD1((kt + 2) * 128, so_cur);
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (TN1 > 0) pp_lgkm_fence<TN1>(bf1);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (!(abl & 2))
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < TM0; ++i)
#pragma unroll
                for (int j = 0; j < TN1; ++j)
                    acc[i][TN0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf1[j][kk]), __builtin_bit_cast(bf16x8_t, af[i][kk]), acc[i][TN0 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 8)) __builtin_amdgcn_s_barrier();
        // ---------------- P2: quadrant (q1, q1); load B(q0) two ahead
        if constexpr (TM1 > 0) { if (!(abl & 4)) PpReadTiles<0, TM1, TM0>::run(af, fa_k0, fa_k1); }
        if constexpr (MODE == PP_PRELAST) {
            if (has_next) { setD2(nxt); issueD2(0, so_cur); }
        } else if constexpr (MODE == PP_LAST) {
            if (has_next) issueD2(128, so_cur);
        } else {
            issueD2((kt + 2) * 128, so_cur);
        }
        __builtin_amdgcn_s_barrier();
        if constexpr (TM1 > 0) pp_lgkm_fence<TM1>(af);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (!(abl & 2))
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < TM1; ++i)
#pragma unroll
                for (int j = 0; j < TN1; ++j)
                    acc[TM0 + i][TN0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf1[j][kk]), __builtin_bit_cast(bf16x8_t, af[i][kk]), acc[TM0 + i][TN0 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 8)) __builtin_amdgcn_s_barrier();
        // ---------------- P3: quadrant (q1, q0): no reads (the B(q0) fragments are still in registers); load A(own, q1) two ahead
        if constexpr (MODE == PP_PRELAST) {
            if (has_next) { setD3(nxt); issueD3(0, so_cur); }
        } else if constexpr (MODE == PP_LAST) {
            if (has_next) issueD3(128, so_cur);
        } else {
            issueD3((kt + 2) * 128, so_cur);
        }
        if (!(abl & 16)) {
            if constexpr (MODE == PP_FIRST) pp_wait<Cf::W_A, Cf::SC, Cf::SB1, Cf::SB0, Cf::SC + Cf::SB0>(pend);
            else if constexpr (MODE == PP_PRELAST || MODE == PP_LAST) { if (has_next) pp_vmcnt<Cf::W_A>(); else pp_vmcnt<0>(); }
            else pp_vmcnt<Cf::W_A>();
        }
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (!(abl & 2))
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < TM1; ++i)
#pragma unroll
                for (int j = 0; j < TN0; ++j)
                    acc[TM0 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf0[j][kk]), __builtin_bit_cast(bf16x8_t, af[i][kk]), acc[TM0 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        // the last barrier of an output tile's last K-tile: group 0 passes it here, group 1 only after ITS epilogue (below, in the tile
        // loop) -- so both groups write their halves of the tile out in the SAME barrier interval, side by side, instead of one after the
        // other (each group's epilogue used to hold the other group at a barrier: 2 x 4.4 us per 160 x 256 tile)
        if (MODE != PP_LAST || !g1) { if (!(abl & 8)) __builtin_amdgcn_s_barrier(); }
        fa_k0 += sdelta; fa_k1 += sdelta; fb_k0 += sdelta; fb_k1 += sdelta;
        sdelta = -sdelta;
        { const int t = so_cur; so_cur = so_oth; so_oth = t; }
    };

    int pend = 0;                                                 // store instructions of the last epilogue still inside the wait windows
    for (int r = 0; r < my_tiles; ++r) {
        const int nxt = r + 1 < my_tiles ? r + 1 : -1;
        ktile(PpMode<PP_FIRST>{}, 0, nxt, pend);
        for (int kt = 1; kt < nkt - 2; ++kt) ktile(PpMode<PP_STEADY>{}, kt, nxt, 0);
        ktile(PpMode<PP_PRELAST>{}, nkt - 2, nxt, 0);
        ktile(PpMode<PP_LAST>{}, nkt - 1, nxt, 0);
        // ---------------- end of an output tile: write it out of the registers, start the next one from zero
        mark();
        int m0, n0;
        tile_mn(r, m0, n0);
        const int lane = pp_lane(), lr = lane & 15, lq = lane >> 4;
        const int mw = m0 + wr * 16 * TM, nw = n0 + wc * 16 * TN;
        const bool whole = m0 + BM <= g.M && n0 + BN <= g.N;
        if (abl & 32) {
            pend = 0;
        } else if (LDS_EPI || nxt < 0) {
            // through the transposing buffers: beyond the ring (one workgroup per CU), or IN the ring once it is dead -- the last tile of a
            // workgroup: every fragment read of its last K-tile retired before the barriers both groups have passed, nothing is in flight
            const unsigned cbuf = lds0 + (LDS_EPI ? Cf::LDS : 0) + wave * 4096;
            if (whole) {
                pp_epilogue_lds<TM, TN, true, OCC == 2>(g, acc, cbuf, mw, nw, lane, pm, dseed);
                pend = (g.C ? Cf::SC : 0) + (g.Cb ? Cf::SB0 : 0);
            } else {
                pp_epilogue_lds<TM, TN, false, OCC == 2>(g, acc, cbuf, mw, nw, lane, pm, dseed);
                pend = 1000;                                      // per-lane tests: the number of store instructions is not known -> drain once
            }
        } else if (whole) {
            pp_epilogue<TM, TN, true>(g, acc, mw, nw, lr, lq, pm, dseed);
            pend = (g.C ? Cf::SC : 0) + (g.Cb ? (pm == 1 ? Cf::SB1 : Cf::SB0) : 0);
        } else {
            pp_epilogue<TM, TN, false>(g, acc, mw, nw, lr, lq, pm, dseed);
            pend = 1000;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        if (g1) { if (!(abl & 8)) __builtin_amdgcn_s_barrier(); }   // group 1's last barrier of the tile (see the end of ktile)
        mark();
    }
    if (!g1) __builtin_amdgcn_s_barrier();                        // group 0 meets group 1's last barrier
    if constexpr (DBG) {
        if (q.stamps && lid < 2 && (wave & 3) == 0 && pp_lane() == 0) {
            unsigned long long rt1;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt1)::"memory");
            unsigned long long* o = q.stamps + (lid * 2 + wr) * 16;
#pragma unroll
            for (int i = 0; i < 12; ++i) o[i] = stamp[i];
            o[12] = (unsigned long long)nstamp; o[13] = rt0; o[14] = rt1;
        }
    }
}

template <int TM, int TN, int OCC>
int pp_launch(const BGemmArgs& g, const PpGeom& q, int grid, hipStream_t st) {
    using Cf = PpCfg<TM, TN>;
    static_assert(OCC == 1 || Cf::LDS <= 80 * 1024, "two workgroups per CU: 80 KB of LDS each");
    constexpr int LDS = Cf::LDS + (OCC == 1 ? Cf::CBUF : 0);
    static CstPerDevice attr_done;
    if (cst_first_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)cst_gemm_bf16_pp_kernel<TM, TN, OCC>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
#ifdef CST_BENCH_VARIANTS
    if constexpr (OCC == 1 && ((TM == 8 && TN == 4) || (TM == 7 && TN == 3) || (TM == 5 && TN == 4) || (TM == 5 && TN == 3))) {
        if (q.abl) {                                              // timing ablations (wrong results): bench build only
            (void)hipFuncSetAttribute((const void*)cst_gemm_bf16_pp_kernel<TM, TN, 1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            hipLaunchKernelGGL((cst_gemm_bf16_pp_kernel<TM, TN, 1, 2>), dim3(grid), dim3(512), LDS, st, g, q);
            return 1;
        }
        if (q.stamps) {                                           // cycle stamps at the section boundaries
            (void)hipFuncSetAttribute((const void*)cst_gemm_bf16_pp_kernel<TM, TN, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
            hipLaunchKernelGGL((cst_gemm_bf16_pp_kernel<TM, TN, 1, 1>), dim3(grid), dim3(512), LDS, st, g, q);
            return 1;
        }
    }
#endif
    if (cst_prof_on()) {
        hipEvent_t ea, eb;
        (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
        cst_prof_push_shape(ea, eb, 2.0 * g.M * g.N * g.K, 2.0 * ((double)g.M * g.K + (double)g.N * g.K) + (g.C ? 4.0 : 0.0) * g.M * g.N + (g.Cb ? 2.0 : 0.0) * g.M * g.N, 2,
                            g.M, g.N, g.K);          // which = 2: cst_gemm_bf16_pp_kernel
        hipExtLaunchKernelGGL((cst_gemm_bf16_pp_kernel<TM, TN, OCC>), dim3(grid), dim3(512), LDS, st, ea, eb, 0, g, q);
    } else {
        hipLaunchKernelGGL((cst_gemm_bf16_pp_kernel<TM, TN, OCC>), dim3(grid), dim3(512), LDS, st, g, q);
    }
    return 1;
}

struct PpChoice { int tm, tn; };
// the instantiated wave tiles (TM x TN 16 x 16 tiles per wave; workgroup tile 32 TM x 64 TN)
constexpr PpChoice PP_MENU[] = {{8, 4}, {7, 4}, {6, 4}, {5, 4}, {4, 4}, {8, 3}, {7, 3}, {6, 3}, {5, 3}, {8, 2}, {7, 2}, {6, 2}};

int pp_ncu() {
    static int n[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (n[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n[dev] = v;
    }
    return n[dev];
}

// modelled time of a product on wave tile (tm, tn), in units of one 16 x 16 x 64 MFMA pair per wave: rounds x (K-tiles x work of a
// tile + what a tile costs besides its MFMAs).  The per-tile and per-K-tile overheads are fitted to tools/gemm_pp_bench.py.
double pp_cost(int M, int N, int K, int tm, int tn, int ncu) {
    const long tiles = (long)cst_div_up(M, 32 * tm) * cst_div_up(N, 64 * tn);
    const long rounds = (tiles + ncu - 1) / ncu;
    const double per_k = tm * tn + 0.6 * (tm + tn) + 3.0;       // MFMA pairs + fragment reads / DMA / barriers that do not hide
    const double per_tile = 2.5 * tm * tn + 20.0;                // epilogue + the ring's refill after it
    return (double)rounds * ((K / 64) * per_k + per_tile);
}

}  // namespace

extern "C" int cst_gemm_bf16_pp_config(int M, int N, int K) {
    // 100 tm + tn of the wave tile bgemm_pp_try would pick for an eligible product of this shape, 0 if it leaves the shape to the tile kernels
    if (M < 1024 || N < 256 || K < 192 || K % 64 != 0 || N % 4 != 0) return 0;
    const int ncu = pp_ncu();
    double best = 1e300;
    int pick = 0;
    for (const PpChoice& c : PP_MENU) {
        const long tiles = (long)cst_div_up(M, 32 * c.tm) * cst_div_up(N, 64 * c.tn);
        if (tiles < ncu / 2) continue;                            // too few tiles to fill the chip: the split-K tile kernels do better
        const double t = pp_cost(M, N, K, c.tm, c.tn, ncu);
        if (t < best) { best = t; pick = 100 * c.tm + c.tn; }
    }
    return pick;
}

#ifdef CST_BENCH_VARIANTS
static unsigned long long* g_pp_stamps = nullptr;
// bench build: the last stamped launch's 4 x 16 words (workgroups 0, 1 x wave groups 0, 1): [0..11] s_memtime at the section boundaries,
// [12] boundaries passed, [13], [14] s_memrealtime (100 MHz) at start and end
extern "C" int cst_gemm_bf16_pp_stamps(unsigned long long* out64) {
    if (!g_pp_stamps) return 1;
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    return hipMemcpy(out64, g_pp_stamps, 4 * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 2;
}
#endif

// cfg = 100 TM + TN, + 10000: the two-workgroups-per-CU build of the wave tile, + 20000: one tile per workgroup (grid = tiles) instead of
// persistent workgroups.  -> 1 launched, 0 no such build
static int pp_dispatch(const BGemmArgs& g, int cfg, hipStream_t st) {
    const bool one_tile = cfg >= 20000;
    if (one_tile) cfg -= 20000;
    const int occ = cfg >= 10000 ? 2 : 1;
    if (occ == 2) cfg -= 10000;
    const int tm = cfg / 100, tn = cfg % 100;
    if (tm < 2 || tn < 2) return 0;
    PpGeom q;
    q.tilesM = cst_div_up(g.M, 32 * tm); q.tilesN = cst_div_up(g.N, 64 * tn);
    q.ntiles = q.tilesM * q.tilesN; q.nkt = g.K / 64;
    q.gn = g.gn > 0 ? g.gn : 8;
    if (q.gn > q.tilesN) q.gn = q.tilesN;
    q.pm = (g.Cb && !g.C && g.N % 8 == 0) ? 1 : 0;               // bf16-only output: pair order (used by the direct epilogue of the 128-register builds)
    q.abl = 0;
    q.stamps = nullptr;
#ifdef CST_BENCH_VARIANTS
    { const char* e = getenv("CST_PP_ABL"); if (e) q.abl = atoi(e); }
    if (getenv("CST_PP_STAMPS")) {
        if (!g_pp_stamps) (void)hipMalloc(&g_pp_stamps, 4 * 16 * sizeof(unsigned long long));
        q.stamps = g_pp_stamps;
    }
#endif
    const int slots = occ * pp_ncu();
    const int grid = (one_tile || q.ntiles < slots) ? q.ntiles : slots;
#define PP_CASE(TM_, TN_) if (occ == 1 && tm == TM_ && tn == TN_) return pp_launch<TM_, TN_, 1>(g, q, grid, st)
#define PP_CASE2(TM_, TN_) if (occ == 2 && tm == TM_ && tn == TN_) return pp_launch<TM_, TN_, 2>(g, q, grid, st)
    PP_CASE(8, 4); PP_CASE(7, 4); PP_CASE(6, 4); PP_CASE(5, 4); PP_CASE(4, 4);
    PP_CASE(8, 3); PP_CASE(7, 3); PP_CASE(6, 3); PP_CASE(5, 3);
    PP_CASE(8, 2); PP_CASE(7, 2); PP_CASE(6, 2);
    PP_CASE2(4, 3); PP_CASE2(6, 2); PP_CASE2(5, 2); PP_CASE2(4, 2); PP_CASE2(3, 3);
#undef PP_CASE2
#undef PP_CASE
    return 0;
}

// ---- which build runs a shape: measured once per (shape, output kind) and process.  Every build of this kernel -- and the tile kernels
// without split-K -- sums a dot product in the same order (K-tiles in sequence, two v_mfma_f32_16x16x32_bf16 per K-tile), so the choice
// changes the time, never a bit of the result: replicas of a data-parallel run may pick differently and still agree.
#include <mutex>
#include <unordered_map>
namespace {
std::mutex g_pp_mu;
std::unordered_map<unsigned long long, int> g_pp_pick;            // key -> cfg code (0: leave the shape to the tile kernels)
constexpr PpChoice PP_MENU2[] = {{4, 3}, {6, 2}, {5, 2}, {4, 2}, {3, 3}};

unsigned long long pp_key(const BGemmArgs& g) {
    const unsigned long long kind = (g.C ? 1u : 0u) | (g.Cb ? 2u : 0u) | ((g.addend || g.act >= 3) ? 4u : 0u);
    return ((unsigned long long)g.M << 40) ^ ((unsigned long long)g.N << 22) ^ ((unsigned long long)g.K << 4) ^ kind;
}

int pp_candidates(const BGemmArgs& g, int* out, int cap) {
    const int ncu = pp_ncu();
    int n = 0;
    for (const PpChoice& c : PP_MENU) {
        const long tiles = (long)cst_div_up(g.M, 32 * c.tm) * cst_div_up(g.N, 64 * c.tn);
        if (tiles >= ncu / 2 && n < cap) out[n++] = 100 * c.tm + c.tn;
    }
    for (const PpChoice& c : PP_MENU2) {
        const long tiles = (long)cst_div_up(g.M, 32 * c.tm) * cst_div_up(g.N, 64 * c.tn);
        if (tiles >= ncu && n + 1 < cap) { out[n++] = 10000 + 100 * c.tm + c.tn; out[n++] = 30000 + 100 * c.tm + c.tn; }
    }
    return n;
}

// time every candidate build on the caller's own operands (the product is idempotent: no accumulation, the dropout mask is a function of
// the seed) and remember the fastest.  Only ever on a stream that is not capturing: it synchronises.
int pp_tune(const BGemmArgs& g, hipStream_t st, int (*tiles)(void*), void* ctx) {
    int cand[40];
    const int n = pp_candidates(g, cand, 40);
    if (n == 0) return 0;
    hipEvent_t ea, eb;
    if (hipEventCreate(&ea) != hipSuccess || hipEventCreate(&eb) != hipSuccess) return cst_gemm_bf16_pp_config(g.M, g.N, g.K);
    // 1 untimed + 4 timed launches a candidate, best of two such passes; a build must beat the tile kernels by 3 % to displace them (two
    // near-equal candidates would otherwise flip with the noise of a 40 us measurement from run to run)
    auto time4 = [&](auto&& launch) -> float {
        float best = 1e30f;
        for (int pass = 0; pass < 2; ++pass) {
            (void)hipEventRecord(ea, st);
            for (int k = 0; k < 4; ++k) launch();
            (void)hipEventRecord(eb, st);
            float ms = 0.f;
            if (hipEventSynchronize(eb) != hipSuccess || hipEventElapsedTime(&ms, ea, eb) != hipSuccess) { (void)hipGetLastError(); return 1e30f; }
            if (ms < best) best = ms;
        }
        return best;
    };
    int best = 0;
    float best_ms = 1e30f;
    if (tiles && tiles(ctx) == CST_OK) best_ms = 0.97f * time4([&]() { (void)tiles(ctx); });     // candidate 0: the LDS-DMA tile kernels (their own plan)
    for (int i = 0; i < n; ++i) {
        if (pp_dispatch(g, cand[i], st) != 1) continue;           // also the build's first launch on this device (attribute, code load)
        const float ms = time4([&]() { (void)pp_dispatch(g, cand[i], st); });
        if (ms < best_ms) { best_ms = ms; best = cand[i]; }
    }
    (void)hipEventDestroy(ea); (void)hipEventDestroy(eb);
    if (best > 0) (void)pp_dispatch(g, best, st);                 // (the last candidate timed may have been any build: same result, but keep the
    else if (tiles) (void)tiles(ctx);                             //  launch the caller sees the winner's -- it is what a profile of this call shows)
    if (getenv("CST_GEMM_PP_VERBOSE")) fprintf(stderr, "[cst] gemm_pp tuned %d x %d x %d (out %s%s): cfg %d, %.1f us\n", g.M, g.N, g.K, g.C ? "f32" : "", g.Cb ? "bf16" : "", best, best_ms * 250.f);
    return best;
}
}  // namespace

// CST_GEMM_PP_CACHE=<file>: the measured choices are read from it when the first product arrives and appended to it as shapes are measured,
// so a second process (a profiled run, a production job) starts with them and measures nothing ("M N K kind cfg" per line, this device class
// only -- the file is the caller's to keep per machine type).
namespace {
bool g_pp_cache_loaded = false;
void pp_cache_load() {                                            // g_pp_mu held
    if (g_pp_cache_loaded) return;
    g_pp_cache_loaded = true;
    const char* path = getenv("CST_GEMM_PP_CACHE");
    if (!path) return;
    if (FILE* f = fopen(path, "r")) {
        unsigned long long key;
        int cfg;
        while (fscanf(f, "%llu %d", &key, &cfg) == 2) g_pp_pick[key] = cfg;
        fclose(f);
    }
}
void pp_cache_append(unsigned long long key, int cfg) {
    const char* path = getenv("CST_GEMM_PP_CACHE");
    if (!path) return;
    if (FILE* f = fopen(path, "a")) { fprintf(f, "%llu %d\n", key, cfg); fclose(f); }
}
}  // namespace

int bgemm_pp_try(const BGemmArgs& g, int force_cfg, hipStream_t st, int (*tiles)(void*), void* ctx) {
    static const int mode = getenv("CST_GEMM_PP") ? atoi(getenv("CST_GEMM_PP")) : 1;   // 0: never (A/B switch), 1: measured choice, 2: the model's choice only
    if (mode == 0 && force_cfg <= 0) return 0;
    if (g.A2 || g.B2 || g.slab_only || g.splits > 1 || g.amax || g.bscale) return 0;
    if (g.K % 64 != 0 || g.K < 192 || g.N % 4 != 0) return 0;           // at least three K-tiles per output tile (first / second to last / last)
    // vector epilogue: 16-byte fp32 pieces, 8- or 16-byte bf16 pieces
    auto al = [](const void* p, unsigned a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
    if (g.C && (g.ldc % 4 != 0 || !al(g.C, 16))) return 0;
    if (g.Cb && (g.ldcb % 8 != 0 || !al(g.Cb, 16))) return 0;
    if (g.addend && (g.ldadd % 4 != 0 || !al(g.addend, 16))) return 0;
    if (g.act >= 3 && (g.ldaux % 4 != 0 || !al(g.aux, 8))) return 0;
    if (g.bias && !al(g.bias, 16)) return 0;
    if (force_cfg > 0) return pp_dispatch(g, force_cfg, st);
    if (cst_gemm_bf16_pp_config(g.M, g.N, g.K) <= 0) return 0;          // shapes the model never gives to this kernel
    {   // A/B aid (tools/gemm_pp_instep.sh): one build for every eligible product of the process
        static const int force_all = getenv("CST_GEMM_PP_FORCE") ? atoi(getenv("CST_GEMM_PP_FORCE")) : 0;
        if (force_all > 0) { const int rc = pp_dispatch(g, force_all, st); if (rc == 1) return 1; }
    }
    int cfg = -1;
    const unsigned long long key = pp_key(g);
    {
        std::lock_guard<std::mutex> lk(g_pp_mu);
        pp_cache_load();
        auto it = g_pp_pick.find(key);
        if (it != g_pp_pick.end()) cfg = it->second;
    }
    if (cfg < 0) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        const bool capturing = hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone;
        if (mode == 1 && !capturing && !cst_prof_on() && !(g.C && g.accumulate)) {
            cfg = pp_tune(g, st, tiles, ctx);
            std::lock_guard<std::mutex> lk(g_pp_mu);
            g_pp_pick[key] = cfg;
            pp_cache_append(key, cfg);
            return (cfg > 0 || tiles) ? 1 : 0;                    // every candidate computed the product; the winner ran last
        } else {
            cfg = cst_gemm_bf16_pp_config(g.M, g.N, g.K);          // not measurable here (capture, profiling run, C += ...): the model, not remembered
        }
    }
    return cfg > 0 ? pp_dispatch(g, cfg, st) : 0;
}
