// EXPERIMENT (round 3, tools/gemm_rs_bench.py): the vendor library's structure for the encoder-layer products -- one 4-wave workgroup per
// CU, 256 x 128 macro tile, 128 x 64 wave tiles (128 accumulator registers a lane), operands staged through VGPRs
// (global_load_dwordx4 -> ds_write_b128, PF K-tiles in flight in registers) into a double-buffered LDS image -- written to find out
// whether that structure beats cst_gemm_bf16_kernel's many small LDS-DMA tiles on 9216 x 768 x 2048 and friends (DESIGN.md section 6,
// "calibration against the vendor library").  bf16 in, bf16 out, no epilogue options, M % 256 == N % 128 == K % 64 == 0.
#include "cst_common.h"
#include <type_traits>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;

namespace {

constexpr int RS_BM = 256, RS_BN = 128;
constexpr int RS_STAGE = (RS_BM + RS_BN) * 128;            // one K-tile image: rows of 128 bytes (64 bf16), XOR-swizzled 16-byte slots

__device__ __forceinline__ int rs_off(int row, int slot) { return row * 128 + ((slot ^ (row & 7)) << 4); }
__device__ __forceinline__ unsigned rs_pack2(float a, float b) {
    __bf16 x = (__bf16)a, y = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, x) | ((unsigned)__builtin_bit_cast(unsigned short, y) << 16);
}

struct RsArgs { const bf16_t* A; long lda; const bf16_t* B; long ldb; bf16_t* C; long ldc; int M, N, K; };

template <int PF, int NK>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void gemm_rs_kernel(RsArgs g) {
    extern __shared__ __attribute__((aligned(16))) char rs_smem[];
    const int tilesN = g.N / RS_BN;
    int id;
    {   // consecutive workgroup ids go round-robin over the 8 XCDs: give each XCD a contiguous range of tiles
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int tm = id / tilesN, tn = id - tm * tilesN;
    const int m0 = tm * RS_BM, n0 = tn * RS_BN;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, lq = lane >> 4;
    const int lrow = lane >> 3, lps = lane & 7;

    // one load instruction = 8 rows x 128 B; wave w stages A chunks 8w .. 8w + 7 and B chunks 4w .. 4w + 3 of every K-tile
    const bf16_t* ap[8];
    const bf16_t* bp[4];
    int aw[8], bw[4];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int r = (wave * 8 + c) * 8 + lrow;
        ap[c] = g.A + (long)(m0 + r) * g.lda + lps * 8;
        aw[c] = rs_off(r, lps);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int r = (wave * 4 + c) * 8 + lrow;
        bp[c] = g.B + (long)(n0 + r) * g.ldb + lps * 8;
        bw[c] = RS_BM * 128 + rs_off(r, lps);
    }
    constexpr int nk = NK;                                 // the K loop is straight-line code: hipcc then counts vmcnt exactly across K-tiles
    u32x4_t ra[PF][8], rb[PF][4];
    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    auto issue = [&](int t, u32x4_t (&xa)[8], u32x4_t (&xb)[4]) {
        const int k = t * 64;
#pragma unroll
        for (int c = 0; c < 8; ++c) xa[c] = *reinterpret_cast<const u32x4_t*>(ap[c] + k);
#pragma unroll
        for (int c = 0; c < 4; ++c) xb[c] = *reinterpret_cast<const u32x4_t*>(bp[c] + k);
    };
    auto stage = [&](int t, const u32x4_t (&xa)[8], const u32x4_t (&xb)[4]) {
        char* st = rs_smem + (t & 1) * RS_STAGE;
#pragma unroll
        for (int c = 0; c < 8; ++c) *reinterpret_cast<u32x4_t*>(st + aw[c]) = xa[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) *reinterpret_cast<u32x4_t*>(st + bw[c]) = xb[c];
    };
    // fragment sets of the two 32-wide halves of a K-tile: each is requested 32 MFMAs before its first use (one wave per SIMD: nothing
    // else hides the LDS latency); sched_barriers pin the phases, hipcc counts lgkmcnt itself
    u32x4_t fa[2][8], fb[2][4];
    auto read_half = [&](int t, int kk, u32x4_t (&af)[8], u32x4_t (&bf)[4]) {
        const char* As = rs_smem + (t & 1) * RS_STAGE;
        const char* Bs = As + RS_BM * 128;
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const u32x4_t*>(Bs + rs_off(wn * 64 + j * 16 + lr, kk * 4 + lq));
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const u32x4_t*>(As + rs_off(wm * 128 + i * 16 + lr, kk * 4 + lq));
    };
    auto mfma_half = [&](const u32x4_t (&af)[8], const u32x4_t (&bf)[4]) {
        // operands swapped: the 16 x 16 result is C^T, so a lane holds four CONSECUTIVE columns n = 4 lq + r of row m = lr
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, bf[j]), __builtin_bit_cast(bf16x8_t, af[i]), acc[i][j], 0, 0, 0);
    };

    // prologue: PF K-tiles requested, the first one staged, its first half's fragments requested
#pragma unroll
    for (int p = 0; p < PF; ++p)
        if (p < nk) issue(p, ra[p], rb[p]);
    stage(0, ra[0], rb[0]);
    __syncthreads();
    read_half(0, 0, fa[0], fb[0]);

    // step for K-tile t, S = t % PF (compile time): set S is free (tile t was staged in the step before), set (S + 1) % PF holds tile t + 1
    // One wave per SIMD: the only thing that can issue under an MFMA's 16 cycles is this wave's own memory instructions, so each half
    // is ONE scheduling region whose order is pinned to MFMA, memory op, MFMA, memory op ... (sched_group_barrier: 0x008 MFMA, 0x200 DS
    // write, 0x100 DS read, 0x020 VMEM read).  First half: stage tile t + 1 (12 writes) and request the second half's fragments
    // (12 reads) under 32 MFMAs; second half: request K-tile t + PF (12 loads) and the next tile's first fragments (12 reads).
    auto step = [&](int t, auto S_) {
        constexpr int S = decltype(S_)::value, S1 = (S + 1) % PF;
        __builtin_amdgcn_sched_barrier(0);
        // (reads before writes in program order: hipcc cannot tell the two LDS stages apart, so whichever comes second waits for all of the first)
        read_half(t, 1, fa[1], fb[1]);
        if (t + 1 < nk) stage(t + 1, ra[S1], rb[S1]);     // hipcc waits for exactly these registers (newer K-tiles stay in flight)
        mfma_half(fa[0], fb[0]);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();                                   // tile t + 1 staged by every wave, tile t read by every wave
        if (t + PF < nk) issue(t + PF, ra[S], rb[S]);
        if (t + 1 < nk) read_half(t + 1, 0, fa[0], fb[0]);
        mfma_half(fa[1], fb[1]);
#pragma unroll
        for (int q = 0; q < 12; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll
    for (int t = 0; t < nk; t += PF) {
        step(t, std::integral_constant<int, 0>{});
        if (PF > 1 && t + 1 < nk) step(t + 1, std::integral_constant<int, 1 % PF>{});
        if (PF > 2 && t + 2 < nk) step(t + 2, std::integral_constant<int, 2 % PF>{});
    }
    __syncthreads();

    // C (bf16) through LDS: [256 rows][128 columns] bf16, 256-byte rows, 16-byte slots XOR-swizzled by the row
    char* cs = rs_smem;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = wm * 128 + i * 16 + lr, col = wn * 64 + j * 16 + 4 * lq;        // 4 columns = 8 bytes
            u32x2_t v;
            v.x = rs_pack2(acc[i][j][0], acc[i][j][1]); v.y = rs_pack2(acc[i][j][2], acc[i][j][3]);
            *reinterpret_cast<u32x2_t*>(cs + row * 256 + ((((col >> 3) ^ (row & 15)) << 4) | ((col & 4) << 1))) = v;
        }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int idx = p * 256 + threadIdx.x;             // 16-byte piece: row idx / 16, slot idx % 16
        const int row = idx >> 4, sl = idx & 15;
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(cs + row * 256 + ((sl ^ (row & 15)) << 4));
        *reinterpret_cast<u32x4_t*>(g.C + (long)(m0 + row) * g.ldc + n0 + sl * 8) = v;
    }
}

}  // namespace

extern "C" int cst_gemm_bf16_rs(const void* A, long lda, const void* B, long ldb, void* C_bf16, long ldc, int M, int N, int K, int prefetch, void* stream) {
    CST_REQUIRE(A && B && C_bf16, "cst_gemm_bf16_rs: null operand");
    CST_REQUIRE(M > 0 && N > 0 && K > 0 && M % 256 == 0 && N % 128 == 0 && K % 64 == 0, "cst_gemm_bf16_rs: needs M %% 256 == N %% 128 == K %% 64 == 0 (M=%d N=%d K=%d)", M, N, K);
    CST_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 &&
                ((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C_bf16)) & 15) == 0, "cst_gemm_bf16_rs: operands must be 16-byte aligned");
    RsArgs g{(const bf16_t*)A, lda, (const bf16_t*)B, ldb, (bf16_t*)C_bf16, ldc, M, N, K};
    const int tiles = (M / RS_BM) * (N / RS_BN);
    const size_t lds = 2 * RS_STAGE;
    hipStream_t st = (hipStream_t)stream;
#define RS_LAUNCH(PF_, NK_) do { \
        (void)hipFuncSetAttribute((const void*)gemm_rs_kernel<PF_, NK_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((gemm_rs_kernel<PF_, NK_>), dim3(tiles), dim3(256), lds, st, g); } while (0)
    const int nk = K / 64;
    if (prefetch == 2) {
        if (nk == 12) RS_LAUNCH(2, 12); else if (nk == 32) RS_LAUNCH(2, 32); else if (nk == 36) RS_LAUNCH(2, 36); else if (nk == 64) RS_LAUNCH(2, 64);
        else { cst_set_error("cst_gemm_bf16_rs: K / 64 must be 12, 32, 36 or 64"); return CST_ERR_ARG; }
    } else {
        if (nk == 12) RS_LAUNCH(3, 12); else if (nk == 32) RS_LAUNCH(3, 32); else if (nk == 36) RS_LAUNCH(3, 36); else if (nk == 64) RS_LAUNCH(3, 64);
        else { cst_set_error("cst_gemm_bf16_rs: K / 64 must be 12, 32, 36 or 64"); return CST_ERR_ARG; }
    }
#undef RS_LAUNCH
    CST_LAUNCH_CHECK("cst_gemm_bf16_rs");
    return CST_OK;
}
