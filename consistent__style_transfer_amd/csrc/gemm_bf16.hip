// bf16-operand "NT" GEMM with direct-to-LDS staging for gfx950, and the cast kernels that feed it.
//
//   C[M,N] = epilogue( alpha * A[M,K] . B[N,K]^T ),  A and B bf16 in HBM with K contiguous and
//   zero-padded to a multiple of 64 (the cast kernels below write that padding).
//
// Why a second GEMM: with fp32 operands in HBM (gemm.hip) every tile goes HBM -> VGPR -> convert ->
// LDS, which costs registers (few waves resident), VALU converts and 2x the L2/HBM bytes, and the
// loads of tile t+1 cannot stay in flight across the barrier without a second register set.  Here
// operands are already bf16 and K-major on both sides (transposed copies are made once by
// cst_cast_bf16 / cst_transpose_bf16, so forward, dgrad and wgrad are all NT products), so tiles go
// HBM -> LDS with global_load_lds_dwordx4 (no VGPR destination) into an NSTAGE-deep ring:
//   iteration t:  s_waitcnt vmcnt(loads of the newer tiles) ; s_barrier ; issue tile t+NSTAGE-1 ;
//                 ds_read_b128 fragments of tile t ; 32 x v_mfma_f32_16x16x32_bf16
// i.e. one raw barrier per K-tile and NSTAGE-1 tiles of loads in flight behind the MFMAs (counted vmcnt,
// never a drain inside the loop; all LDS lives in ONE array so hipcc adds no vmcnt(0) of its own).
// The shipped dispatch instantiates NSTAGE = 2 only (48 KiB at 64 x 128: three workgroups per CU; 3- and 4-deep rings
// were measured slower because they cost a resident workgroup -- tile codes 65/129/130 keep them reachable for benchmarks).
// LDS image = [rows][128 B] with the 16-byte slot XOR-swizzle of gemm.hip; because a
// global_load_lds wave-instruction writes 1 KiB linearly (lane l -> base + 16 l), the swizzle is
// applied to the per-lane SOURCE address (same 128-byte line, so coalescing is unchanged).
//
// Reference call sites served: the packed in_proj / out_proj / linear1 / linear2 products of
// nn.TransformerEncoderLayer (mlm.py:20-22, match.py:18-20), forward, dgrad and wgrad.
#include "cst_common.h"
#include "bgemm.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

__device__ __forceinline__ bf16_t f2bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf162f(bf16_t h) { return __uint_as_float((uint32_t)h << 16); }

// =============================================================================================
// cast / transpose
// =============================================================================================
// out[r][c]   = bf16(x[r][c] * dropmask(r*C + c))   for c < C, 0 for C <= c < ldo   (r < R)
// out_t[c][r] = same value                          for r < R, 0 for R <= r < ldot  (c < C)
// One 64x64 tile per block, transposed through LDS.  IN_BF16: x is already bf16 (pure transpose).
template <bool IN_BF16>
__global__ __launch_bounds__(256) void cast_bf16_kernel(const void* __restrict__ xin, long ldx, int R, int C,
                                                        bf16_t* __restrict__ out, long ldo, int Cp,
                                                        bf16_t* __restrict__ out_t, long ldot, int Rp, CstDrop drop) {
    __shared__ bf16_t tile[64][66];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    // 256 threads: 4 rows per pass, 64 columns
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int rr = i >> 6, cc = i & 63;
        const int r = r0 + rr, c = c0 + cc;
        float v = 0.f;
        if (r < R && c < C) {
            if constexpr (IN_BF16) v = bf162f(reinterpret_cast<const bf16_t*>(xin)[(long)r * ldx + c]);
            else v = reinterpret_cast<const float*>(xin)[(long)r * ldx + c];
            if (drop.p > 0.f) v *= cst_drop_mask(drop, dseed, (uint32_t)((long)r * C + c));
        }
        const bf16_t h = f2bf16(v);
        tile[rr][cc] = h;
        if (out && r < R && c < Cp) out[(long)r * ldo + c] = h;
    }
    if (!out_t) return;
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int cc = i >> 6, rr = i & 63;          // consecutive threads walk r (contiguous in out_t)
        const int r = r0 + rr, c = c0 + cc;
        if (c < C && r < Rp) out_t[(long)c * ldot + r] = tile[rr][cc];
    }
}

// fp32 input without dropout, everything 4-element aligned (every weight and activation of the module constants): float4
// loads all in flight, 8-byte stores for both images, the transposed one gathered from LDS four rows at a time.
__device__ __forceinline__ void cast_bf16_vec_tile(const float* __restrict__ x, long ldx, int R, int C,
                                                   bf16_t* __restrict__ out, long ldo, int Cp,
                                                   bf16_t* __restrict__ out_t, long ldot, int Rp, const int bx, const int by) {
    __shared__ __attribute__((aligned(8))) bf16_t tile[64][68];
    const int r0 = by * 64, c0 = bx * 64;
    float4 v[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int idx = threadIdx.x + 256 * p;
        const int r = r0 + (idx >> 4), c = c0 + (idx & 15) * 4;
        const bool ok = r < R && c < C;
        const float4* src = reinterpret_cast<const float4*>(x + (long)(ok ? r : 0) * ldx + (ok ? c : 0));
        const float4 t = *src;                                        // unconditional (clamped) load, select after
        v[p] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int idx = threadIdx.x + 256 * p;
        const int rr = idx >> 4, cc = (idx & 15) * 4;
        const int r = r0 + rr, c = c0 + cc;
        uint2 u;
        u.x = (uint32_t)f2bf16(v[p].x) | ((uint32_t)f2bf16(v[p].y) << 16);
        u.y = (uint32_t)f2bf16(v[p].z) | ((uint32_t)f2bf16(v[p].w) << 16);
        *reinterpret_cast<uint2*>(&tile[rr][cc]) = u;
        if (out && r < R && c < Cp) *reinterpret_cast<uint2*>(out + (long)r * ldo + c) = u;
    }
    if (!out_t) return;
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int idx = threadIdx.x + 256 * p;
        const int cc = idx >> 4, rr = (idx & 15) * 4;                 // 16 consecutive lanes walk r (contiguous in out_t)
        const int r = r0 + rr, c = c0 + cc;
        if (c < C && r < Rp) {
            uint2 u;
            u.x = (uint32_t)tile[rr][cc] | ((uint32_t)tile[rr + 1][cc] << 16);
            u.y = (uint32_t)tile[rr + 2][cc] | ((uint32_t)tile[rr + 3][cc] << 16);
            *reinterpret_cast<uint2*>(out_t + (long)c * ldot + r) = u;
        }
    }
}

__global__ __launch_bounds__(256) void cast_bf16_vec_kernel(const float* __restrict__ x, long ldx, int R, int C,
                                                            bf16_t* __restrict__ out, long ldo, int Cp,
                                                            bf16_t* __restrict__ out_t, long ldot, int Rp) {
    cast_bf16_vec_tile(x, ldx, R, C, out, ldo, Cp, out_t, ldot, Rp, blockIdx.x, blockIdx.y);
}

// Many matrices in one launch (cst_cast_bf16_multi: the bf16 twins of every trained weight of an optimizer group, once per optimizer
// step): the jobs ride in the kernel arguments, a block finds its job from the running block counts.
#define CST_CAST_MULTI_MAX 40
struct CastJob { const float* x; bf16_t* out; bf16_t* out_t; int ldx, ldo, ldot, R, C, gx, start; };
struct CastJobs { CastJob j[CST_CAST_MULTI_MAX]; int n; };

__global__ __launch_bounds__(256) void cast_bf16_multi_kernel(CastJobs q) {
    int k = 0;
    for (int i = 1; i < q.n; ++i)
        if ((int)blockIdx.x >= q.j[i].start) k = i;
    const CastJob& j = q.j[k];
    const int local = blockIdx.x - j.start;
    cast_bf16_vec_tile(j.x, j.ldx, j.R, j.C, j.out, j.ldo, j.out ? j.ldo : j.C, j.out_t, j.ldot, j.out_t ? j.ldot : j.R, local % j.gx, local / j.gx);
}

/* table: n rows of 8 host int64 words {x, ldx, R, C, out, ldo, out_t, ldot} -- the arguments of n cst_cast_bf16 calls on fp32 inputs
 * without dropout, every one of which must qualify for the vector kernel (C, ldx, ldo, ldot multiples of 4; x 16-byte, outputs 8-byte aligned) */
extern "C" int cst_cast_bf16_multi(const long* table, int n, void* stream) {
    CST_REQUIRE(table && n > 0, "cst_cast_bf16_multi: empty table");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < n; base += CST_CAST_MULTI_MAX) {
        CastJobs q{};
        q.n = n - base < CST_CAST_MULTI_MAX ? n - base : CST_CAST_MULTI_MAX;
        long blocks = 0;
        for (int i = 0; i < q.n; ++i) {
            const long* t = table + (long)(base + i) * 8;
            const void* x = (const void*)t[0]; const long ldx = t[1]; const long R = t[2], C = t[3];
            void* out = (void*)t[4]; const long ldo = t[5]; void* out_t = (void*)t[6]; const long ldot = t[7];
            CST_REQUIRE(x && (out || out_t) && R > 0 && C > 0 && ldx >= C && (!out || ldo >= C) && (!out_t || ldot >= R) && ldx < (1L << 31) && ldo < (1L << 31) && ldot < (1L << 31),
                        "cst_cast_bf16_multi: job %d: bad arguments", base + i);
            CST_REQUIRE(C % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0 && (!out || (ldo % 4 == 0 && (((uintptr_t)out) & 7) == 0)) &&
                        (!out_t || (ldot % 4 == 0 && (((uintptr_t)out_t) & 7) == 0)), "cst_cast_bf16_multi: job %d does not qualify for the vector kernel", base + i);
            const long Cp = out ? ldo : C, Rp = out_t ? ldot : R;
            const int gx = cst_div_up((int)(Cp > C ? Cp : C), 64), gy = cst_div_up((int)(Rp > R ? Rp : R), 64);
            q.j[i] = CastJob{(const float*)x, (bf16_t*)out, (bf16_t*)out_t, (int)ldx, (int)ldo, (int)ldot, (int)R, (int)C, gx, (int)blocks};
            blocks += (long)gx * gy;
        }
        hipLaunchKernelGGL(cast_bf16_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, st, q);
        CST_LAUNCH_CHECK("cst_cast_bf16_multi");
    }
    return CST_OK;
}

extern "C" int cst_cast_bf16(const void* x, int x_is_bf16, long ldx, int R, int C,
                             void* out, long ldo, void* out_t, long ldot,
                             float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream) {
    CST_REQUIRE(x && (out || out_t) && R > 0 && C > 0 && ldx >= C, "cst_cast_bf16: bad arguments");
    CST_REQUIRE(!out || ldo >= C, "cst_cast_bf16: ldo < C");
    CST_REQUIRE(!out_t || ldot >= R, "cst_cast_bf16: ldot < R");
    // padded extents: every column of `out` up to ldo and every column of `out_t` up to ldot is written
    const int Cp = out ? (int)ldo : C, Rp = out_t ? (int)ldot : R;
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)R * C);
    dim3 grid(cst_div_up(Cp > C ? Cp : C, 64), cst_div_up(Rp > R ? Rp : R, 64)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = !x_is_bf16 && drop_p <= 0.f && C % 4 == 0 && ldx % 4 == 0 && (((uintptr_t)x) & 15) == 0 &&
                     (!out || (ldo % 4 == 0 && (((uintptr_t)out) & 7) == 0)) && (!out_t || (ldot % 4 == 0 && (((uintptr_t)out_t) & 7) == 0));
    if (vec) {
        hipLaunchKernelGGL(cast_bf16_vec_kernel, grid, block, 0, st, (const float*)x, ldx, R, C, (bf16_t*)out, ldo, Cp, (bf16_t*)out_t, ldot, Rp);
        CST_LAUNCH_CHECK("cst_cast_bf16");
        return CST_OK;
    }
    if (x_is_bf16) hipLaunchKernelGGL((cast_bf16_kernel<true>), grid, block, 0, st, x, ldx, R, C, (bf16_t*)out, ldo, Cp, (bf16_t*)out_t, ldot, Rp, dr);
    else hipLaunchKernelGGL((cast_bf16_kernel<false>), grid, block, 0, st, x, ldx, R, C, (bf16_t*)out, ldo, Cp, (bf16_t*)out_t, ldot, Rp, dr);
    CST_LAUNCH_CHECK("cst_cast_bf16");
    return CST_OK;
}

// column sums of a bf16 matrix into fp32 (bias gradients of bf16-only activations gradients)
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ X, long ld, int M, int N,
                                                          float* __restrict__ out, int rows_per_split) {
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const long r0 = (long)blockIdx.y * rows_per_split;
    float s = 0.f;
    if (c < N) {
        const long rend = min((long)M, r0 + rows_per_split);
        float s0 = 0.f, s1 = 0.f;
        long r = r0 + w;
        for (; r + 4 < rend; r += 8) { s0 += bf162f(X[r * ld + c]); s1 += bf162f(X[(r + 4) * ld + c]); }
        for (; r < rend; r += 4) s0 += bf162f(X[r * ld + c]);
        s = s0 + s1;
    }
    sh[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < N) atomicAdd(out + c, (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]));
}

extern "C" int cst_colsum_bf16(const void* X, long ld, int M, int N, float* out, int accumulate, void* stream) {
    CST_REQUIRE(X && out && M > 0 && N > 0 && ld >= N, "cst_colsum_bf16: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int cg = cst_div_up(N, 64);
    int splits = 2048 / cg; if (splits < 1) splits = 1; if (splits > M / 32) splits = M / 32; if (splits < 1) splits = 1;
    // the splits add into `out` with atomics: accumulate != 0 = the caller's `out` already holds what the sums are added to (zeros from its
    // zero arena, or a running gradient)
    if (!accumulate && cst_zero_words(out, N, st) != CST_OK) { cst_set_error("cst_colsum_bf16: zero fill failed"); return CST_ERR_LAUNCH; }
    hipLaunchKernelGGL(colsum_bf16_kernel, dim3(cg, splits), dim3(256), 0, st, (const bf16_t*)X, ld, M, N, out, cst_div_up(M, splits));
    CST_LAUNCH_CHECK("cst_colsum_bf16");
    return CST_OK;
}

// =============================================================================================
// GEMM
// =============================================================================================

// Bench-only code (tools/gemm_bench.py): the timing ablations and the kernel variants that were measured and never dispatch (256-wide
// tiles, the loader / consumer kernel, 3- and 4-stage rings) are compiled only with -DCST_BENCH_VARIANTS (CST_BENCH_VARIANTS=1 python -m
// consistent__style_transfer_amd.build --force); the shipped library holds neither, so a leaked CST_GB_ABL cannot corrupt a run.
#ifdef CST_BENCH_VARIANTS
#define BGEMM_ABL(g) ((g).abl)
#else
#define BGEMM_ABL(g) 0
#endif
extern "C" int cst_bench_variants() {
#ifdef CST_BENCH_VARIANTS
    return 1;
#else
    return 0;
#endif
}

constexpr int BBK = 64;             // bf16 elements of K per tile = 128 bytes per row
constexpr int BROW = 128;

__device__ __forceinline__ int blds_off(int row, int slot) { return row * BROW + ((slot ^ (row & 7)) << 4); }

// Fragment reads are inline asm on purpose: hipcc treats an in-flight global_load_lds as a pending
// LDS write and drains vmcnt(0) in front of any ds_read it can see, which would collapse the
// 3-stage ring to one tile in flight.  These reads are invisible to that pass; their completion is
// waited for by the explicit lgkmcnt(0) + sched_barrier below (cdna_hip_programming.md 5.7, form iii).
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;
__device__ __forceinline__ u32x4_t lds_read128(unsigned addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// Four 16-byte reads 1 KiB apart AND their wait in one statement: the compiler believes an asm's outputs are valid when the statement
// ends, so a separate wait statement leaves it free to copy a not-yet-landed register in between (it did: two of four columns wrong).
__device__ __forceinline__ void lds_read128x4_wait(unsigned addr, u32x4_t (&v)[4]) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                 "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(addr) : "memory");
}

typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
__device__ __forceinline__ u32x2_t lds_read64_tr(unsigned addr) {
    u32x2_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// byte offset of 16-byte chunk ch (0..15) of row `row` in the [64 k][128 x bf16] transposed-read image
__device__ __forceinline__ int tlds_off(int row, int ch) { return row * 256 + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); }

// Arg-max of a row as ONE 64-bit word that an atomic max can maintain in any arrival order: the high half orders like the float
// (sign-flipped bits), the low half is the inverted column index, so among equal values the FIRST column wins -- torch.argmax's answer
// (rnn.py:53, :92).  A zeroed word is below every real entry.
__device__ __forceinline__ unsigned long long bgemm_pack_max(float v, int idx) {
    unsigned u = __float_as_uint(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)idx);
}
template <int CTRL>
__device__ __forceinline__ int cst_dpp_i(int v, int old) { return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false); }
__device__ __forceinline__ int row16_min_i(int v) {
    v = min(v, cst_dpp_i<0xB1>(v, v));
    v = min(v, cst_dpp_i<0x4E>(v, v));
    v = min(v, cst_dpp_i<0x141>(v, v));
    v = min(v, cst_dpp_i<0x140>(v, v));
    return v;
}

// One word per row would put all 2 * N / 128 atomics of a row -- and of its seven neighbours -- on ONE 64-byte line: 40 000 atomics of a
// 256 x 10 000 product on 32 lines, serialised at the memory side (+6 us on a 13 us launch; +3 us with 16 words per row).  So a row owns
// CST_AMAX_GROUPS words, one per group of column tiles (128-column tile index mod 32), group-major [group][M]: the consumer takes the max.
#define CST_AMAX_GROUPS 32
__device__ __forceinline__ unsigned long long* bgemm_amax_word(const BGemmArgs& g, int m, int n) {
    return g.amax + (long)((n >> 7) & (CST_AMAX_GROUPS - 1)) * g.M + m;
}

__device__ __forceinline__ void bgemm_store(const BGemmArgs& g, uint32_t dseed, int m, int n, float acc) {
    if (g.bscale) acc *= g.bscale[n];
    float v = g.alpha * acc + (g.bias ? g.bias[n] : 0.f);
    if (g.addend) v += g.addend[(long)m * g.ldadd + n];
    if (g.act == 1) v = v > 0.f ? v : 0.f;
    else if (g.act == 2) v = v > 0.f ? v : 0.1f * v;
    else if (g.act == 3) v = bf162f(g.aux[(long)m * g.ldaux + n]) > 0.f ? v * g.gate_scale : 0.f;
    else if (g.act == 4) v = bf162f(g.aux[(long)m * g.ldaux + n]) > 0.f ? v : 0.1f * v;
    if (g.drop.p > 0.f) v *= cst_drop_mask(g.drop, dseed, (uint32_t)((long)m * g.N + n));
    if (g.C) { float* cp = g.C + (long)m * g.ldc + n; if (g.accumulate) v += *cp; *cp = v; }
    if (g.Cb) g.Cb[(long)m * g.ldcb + n] = f2bf16(v);
    if (g.amax) atomicMax(bgemm_amax_word(g, m, n), bgemm_pack_max(v, n));
}

// Which of the epilogue's streams can move as 16-byte (fp32) / 8-byte (bf16) vectors: four
// consecutive columns per lane.  Evaluated once per kernel (wave-uniform).
__host__ __device__ __forceinline__ bool bgemm_vec_ok(const BGemmArgs& g) {
    const bool vc = !g.C || ((g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.C) & 15) == 0));
    const bool vb = !g.Cb || ((g.ldcb % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.Cb) & 7) == 0));
    const bool va = !g.addend || ((g.ldadd % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.addend) & 15) == 0));
    const bool vx = g.act < 3 || ((g.ldaux % 4 == 0) && ((reinterpret_cast<uintptr_t>(g.aux) & 7) == 0));
    const bool vbias = !g.bias || ((reinterpret_cast<uintptr_t>(g.bias) & 15) == 0);
    return vc && vb && va && vx && vbias;
}

// columns n..n+3 of row m (n % 4 == 0, n + 3 < N, bgemm_vec_ok): same arithmetic as bgemm_store
__device__ __forceinline__ void bgemm_store4(const BGemmArgs& g, uint32_t dseed, int m, int n, const float av[4]) {
    float o[4];
    float b4[4] = {0.f, 0.f, 0.f, 0.f}, a4[4] = {0.f, 0.f, 0.f, 0.f}, x4[4] = {1.f, 1.f, 1.f, 1.f}, s4[4] = {1.f, 1.f, 1.f, 1.f};
    if (g.bscale) { s4[0] = g.bscale[n]; s4[1] = g.bscale[n + 1]; s4[2] = g.bscale[n + 2]; s4[3] = g.bscale[n + 3]; }
    if (g.bias) { const float4 t = *reinterpret_cast<const float4*>(g.bias + n); b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w; }
    if (g.addend) { const float4 t = *reinterpret_cast<const float4*>(g.addend + (long)m * g.ldadd + n); a4[0] = t.x; a4[1] = t.y; a4[2] = t.z; a4[3] = t.w; }
    if (g.act >= 3) {
        const uint2 u = *reinterpret_cast<const uint2*>(g.aux + (long)m * g.ldaux + n);
        x4[0] = bf162f((bf16_t)(u.x & 0xffffu)); x4[1] = bf162f((bf16_t)(u.x >> 16));
        x4[2] = bf162f((bf16_t)(u.y & 0xffffu)); x4[3] = bf162f((bf16_t)(u.y >> 16));
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = g.alpha * (av[e] * s4[e]) + b4[e];
        v += a4[e];
        if (g.act == 1) v = v > 0.f ? v : 0.f;
        else if (g.act == 2) v = v > 0.f ? v : 0.1f * v;
        else if (g.act == 3) v = x4[e] > 0.f ? v * g.gate_scale : 0.f;
        else if (g.act == 4) v = x4[e] > 0.f ? v : 0.1f * v;
        if (g.drop.p > 0.f) v *= cst_drop_mask(g.drop, dseed, (uint32_t)((long)m * g.N + n + e));
        o[e] = v;
    }
    if (g.C) {
        float4* cp = reinterpret_cast<float4*>(g.C + (long)m * g.ldc + n);
        if (g.accumulate) { const float4 t = *cp; o[0] += t.x; o[1] += t.y; o[2] += t.z; o[3] += t.w; }
        *cp = make_float4(o[0], o[1], o[2], o[3]);
    }
    if (g.Cb) {
        uint2 u;
        u.x = (uint32_t)f2bf16(o[0]) | ((uint32_t)f2bf16(o[1]) << 16);
        u.y = (uint32_t)f2bf16(o[2]) | ((uint32_t)f2bf16(o[3]) << 16);
        *reinterpret_cast<uint2*>(g.Cb + (long)m * g.ldcb + n) = u;
    }
    if (g.amax) {
        float bv = o[0]; int bi = n;
#pragma unroll
        for (int e = 1; e < 4; ++e) if (o[e] > bv) { bv = o[e]; bi = n + e; }
        atomicMax(bgemm_amax_word(g, m, n), bgemm_pack_max(bv, bi));
    }
}

// Wave-level write-out of a staged [ROWS][64] fp32 sub-tile (LDS byte address cs, row stride 256 B) to rows m0.., columns n0..n0+63.
//
// Why this shape (round 2, tools/gemm_bench.py abl): gfx950 counts loads AND stores in vmcnt, in issue order.  The per-segment
// loop this replaces (LDS read -> optional bias / addend / aux loads -> store, all behind runtime tests) made hipcc put
// s_waitcnt vmcnt(0) in front of every segment -- once for the C++ LDS read (it cannot prove an LDS-DMA is not pending), once for
// the loads that follow the previous segment's store -- so each wave paid one full store round trip per 1 KiB segment: the write-out
// alone took 28 us (64 x 128 tiles) to 50 us (256 x 256) of a 54-79 us product.  Here a wave first requests its whole sub-tile from
// LDS (inline asm: invisible to that pass) and every per-element operand of the batch, waits once, then issues all its stores back to
// back.  Per-column operands (bias, fp8 channel scale) are loaded once per lane: a lane keeps the same four columns in every segment.
// Requires bgemm_vec_ok and the sub-tile fully inside the matrix (edge tiles keep the element-wise path).
// LOADS = false is the instantiation for launches without per-element operands (no addend, no aux gate, no accumulate): its loop
// holds no global load, so hipcc has no reason to wait on vmcnt inside it and the stores of a wave never wait for each other.  With
// LOADS = true every batch's loads come after the previous batch's stores in the counter, i.e. one store round trip per batch.
// this lane's four columns' bias and fp8 channel scale (the same in every segment), pinned in registers on return
__device__ __forceinline__ void bgemm_col_operands(const BGemmArgs& g, int n0, int lane, float (&s4)[4], float (&b4)[4]) {
    const int n = n0 + ((lane & 15) << 2);
    s4[0] = s4[1] = s4[2] = s4[3] = 1.f;
    b4[0] = b4[1] = b4[2] = b4[3] = 0.f;
    if (g.bscale) { const float4 t = *reinterpret_cast<const float4*>(g.bscale + n); s4[0] = t.x; s4[1] = t.y; s4[2] = t.z; s4[3] = t.w; }
    if (g.bias) { const float4 t = *reinterpret_cast<const float4*>(g.bias + n); b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w; }
    // their vmcnt wait must not end up inside a store loop, behind stores
    asm volatile("" : "+v"(s4[0]), "+v"(s4[1]), "+v"(s4[2]), "+v"(s4[3]), "+v"(b4[0]), "+v"(b4[1]), "+v"(b4[2]), "+v"(b4[3]));
}

template <int ROWS, bool LOADS>
__device__ __forceinline__ void bgemm_write_rows_t(const BGemmArgs& g, uint32_t dseed, unsigned cs, int m0, int n0, int lane,
                                                   const float (&s4)[4], const float (&b4)[4]) {
    constexpr int NB = 4;                               // segments (4 rows x 256 B each) per batch
    static_assert(ROWS % (4 * NB) == 0, "whole batches");
    const int r4 = lane >> 4, n = n0 + ((lane & 15) << 2);
    unsigned ad0 = cs + r4 * 256 + ((lane & 15) << 4);
    const int act = g.act;
    const float alpha = g.alpha;
    const bool has_add = LOADS && g.addend != nullptr, has_old = LOADS && g.C && g.accumulate, has_aux = LOADS && act >= 3;
    // v > 0 ? v * ps : v * ns   (nz: the negative side is an exact +0, as the element-wise path writes it)
    const float ps = act == 3 ? g.gate_scale : 1.f, ns = (act == 2 || act == 4) ? 0.1f : (act == 0 ? 1.f : 0.f);
    const bool nz = act == 1 || act == 3;
    // per-lane row pointers, advanced by four rows per segment (null streams are never dereferenced)
    float* cp = g.C ? g.C + (long)(m0 + r4) * g.ldc + n : nullptr;
    bf16_t* cbp = g.Cb ? g.Cb + (long)(m0 + r4) * g.ldcb + n : nullptr;
    const float* ap = has_add ? g.addend + (long)(m0 + r4) * g.ldadd + n : nullptr;
    const bf16_t* xp = has_aux ? g.aux + (long)(m0 + r4) * g.ldaux + n : nullptr;
    const long c4 = 4 * g.ldc, cb4 = 4 * g.ldcb, a4s = 4 * g.ldadd, x4s = 4 * g.ldaux;
    uint32_t didx = (uint32_t)((long)(m0 + r4) * g.N + n);
#pragma unroll 1
    for (int b0 = 0; b0 < ROWS / 4; b0 += NB) {
        u32x4_t v[NB];
        float4 ad[LOADS ? NB : 1], old[LOADS ? NB : 1];
        uint2 ax[LOADS ? NB : 1];
        if constexpr (LOADS) {
            if (has_add) {
#pragma unroll
                for (int i = 0; i < NB; ++i) ad[i] = *reinterpret_cast<const float4*>(ap + i * a4s);
                ap += NB * a4s;
            }
            if (has_aux) {
#pragma unroll
                for (int i = 0; i < NB; ++i) ax[i] = *reinterpret_cast<const uint2*>(xp + i * x4s);
                xp += NB * x4s;
            }
            if (has_old) {
#pragma unroll
                for (int i = 0; i < NB; ++i) old[i] = *reinterpret_cast<const float4*>(cp + i * c4);
            }
        }
        static_assert(NB == 4, "lds_read128x4_wait");
        lds_read128x4_wait(ad0, v);
        ad0 += NB * 1024;
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            float o[4] = {__uint_as_float(v[i].x), __uint_as_float(v[i].y), __uint_as_float(v[i].z), __uint_as_float(v[i].w)};
            float x4[4] = {1.f, 1.f, 1.f, 1.f};
            if constexpr (LOADS) {
                if (has_aux) {
                    x4[0] = bf162f((bf16_t)(ax[i].x & 0xffffu)); x4[1] = bf162f((bf16_t)(ax[i].x >> 16));
                    x4[2] = bf162f((bf16_t)(ax[i].y & 0xffffu)); x4[3] = bf162f((bf16_t)(ax[i].y >> 16));
                }
            }
            // activation as two wave-uniform slopes and a gate value (act 0-2: the value itself, act 3-4: aux): no branches per element
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float w = alpha * (o[e] * s4[e]) + b4[e];
                if constexpr (LOADS) { if (has_add) w += (&ad[i].x)[e]; }
                const float gv = has_aux ? x4[e] : w;
                o[e] = gv > 0.f ? w * ps : (nz ? 0.f : w * ns);
            }
            if (g.drop.p > 0.f) {
                const uint32_t di = didx + (uint32_t)(4 * i) * (uint32_t)g.N;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] *= cst_drop_mask(g.drop, dseed, di + e);
            }
            if (cp) {
                if constexpr (LOADS) { if (has_old) { o[0] += old[i].x; o[1] += old[i].y; o[2] += old[i].z; o[3] += old[i].w; } }
                *reinterpret_cast<float4*>(cp + i * c4) = make_float4(o[0], o[1], o[2], o[3]);
            }
            if (cbp) {
                uint2 u;
                u.x = (uint32_t)f2bf16(o[0]) | ((uint32_t)f2bf16(o[1]) << 16);
                u.y = (uint32_t)f2bf16(o[2]) | ((uint32_t)f2bf16(o[3]) << 16);
                *reinterpret_cast<uint2*>(cbp + i * cb4) = u;
            }
            if (g.amax) {
                // the 16 lanes of a DPP row hold the 64 columns of one output row: reduce to (largest value, first column) and
                // fold the piece into the row's packed word (all 64 lanes are active here: the sub-tile is whole)
                float bv = o[0]; int bi = n;
#pragma unroll
                for (int e = 1; e < 4; ++e) if (o[e] > bv) { bv = o[e]; bi = n + e; }
                const float rm = row16_max(bv);
                const int ri = row16_min_i(bv == rm ? bi : 0x7fffffff);
                if ((lane & 15) == 0) atomicMax(bgemm_amax_word(g, m0 + r4 + 4 * (b0 + i), n0), bgemm_pack_max(rm, ri));
            }
        }
        if (cp) cp += NB * c4;
        if (cbp) cbp += NB * cb4;
        didx += (uint32_t)(4 * NB) * (uint32_t)g.N;
    }
}

template <int ROWS>
__device__ __forceinline__ void bgemm_write_rows(const BGemmArgs& g, uint32_t dseed, unsigned cs, int m0, int n0, int lane,
                                                 const float (&s4)[4], const float (&b4)[4]) {
    if (!g.addend && g.act < 3 && !(g.C && g.accumulate)) bgemm_write_rows_t<ROWS, false>(g, dseed, cs, m0, n0, lane, s4, b4);
    else bgemm_write_rows_t<ROWS, true>(g, dseed, cs, m0, n0, lane, s4, b4);
}

// TT = false: C = A B^T, A [M,K] and B [N,K] with K contiguous (row reads of the tile: ds_read_b128).
// TT = true : C = A^T B, A [K,M] and B [K,N] with the CONTRACTION index as the row index (weight gradients
//             dW = dY^T X straight from the row-major activations, no transposed copies in HBM): a k-tile is
//             64 rows of 256 bytes and the MFMA fragments come from the hardware transposed read
//             ds_read_b64_tr_b16 (4 rows x 16 columns per 16-lane group, delivered column-major), on the
//             XOR image (b) of cdna_hip_programming.md T10.  128x128 tiles only.
// BF8 = true: the B operand is fp8 e4m3 (OCP) in HBM, [N][K] with K contiguous, one byte per element -- the W8A16 path of
//             BASELINE configs[4] ("fp8 weight MFMA + bf16 activations"): the weight tile travels HBM -> LDS at half the bytes
//             (64-byte rows; 16-row DMA pieces; slot swizzle g((row >> 2) & 3) as in the 256-wide kernel below), each lane reads its
//             8 k-values with one ds_read_b64 and widens them in registers (v_cvt_pk_f32_fp8 x 4, v_cvt_pk_bf16_f32 x 4: e4m3
//             is exactly representable in bf16) for the same v_mfma_f32_16x16x32_bf16; the per-output-channel scale rides in
//             the epilogue (bscale).  gfx950 has no MFMA that mixes bf16 and fp8 operands.
// NW = waves per workgroup: 4 (2 x 2 wave tiles of BM/2 x BN/2) or 8 (4 x 2 wave tiles of BM/4 x BN/2).  128 x 128 / 8 waves (tile code 136)
// keeps the 128 x 128 tile's bytes per FLOP with sixteen DMA-issuing waves per CU instead of eight.
// The body of one workgroup: `id` = its position in the XCD-ordered list of the product's output tiles, `by` = its K split, `bz` = which of
// the two same-shape problems (A2 / B2).  Shared by the one-product kernel and the grouped weight-gradient kernel below.
// SLAB_THROUGH: the partial tile goes to the slab with device-scope stores (written through the XCD's L2), for a reader in ANOTHER workgroup
// of the same launch (grouped weight gradients); otherwise plain stores, made visible by the end of the kernel.
template <int BM, int BN, int NSTAGE, bool TT, bool BF8, int NW, bool SLAB_THROUGH = false>
__device__ __forceinline__ void bgemm_tile_body(const BGemmArgs& g, const int id, const int by, const int bz) {
    static_assert(!(TT && BF8), "fp8 B operand: NT products only");
    constexpr int A_BYTES = BM * BROW, B_BYTES = BN * (BF8 ? 64 : BROW), ST_BYTES = A_BYTES + B_BYTES;
    constexpr int A_CH = BM / 8 / NW, B_CH = BF8 ? BN / 16 / NW : BN / 8 / NW;   // 1-KiB chunks (8 rows; fp8 B: 16 rows) per wave per tile
    constexpr int LOADS = A_CH + B_CH;                          // global_load_lds instructions per wave per tile
    constexpr int WMR = BM / (NW / 2);                         // rows of a wave tile
    constexpr int TM = WMR / 16, TN = BN / 32;
    static_assert(NW == 4 || (NW == 8 && !TT && !BF8), "8 waves: NT products on bf16 operands only");
    static_assert(!TT || (BM == 128 && BN == 128), "the transposed-read image assumes 256-byte tile rows");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    // an XCD (consecutive ids) works on a strip GN tile columns wide: every A panel is then fetched by tilesN / GN XCDs instead of all
    // eight, while the strip's B panels (GN x BN rows of K) stay in its L2.  Sweep (tools/gemm_bench.py enc, CST_GEMM_GN = 2/4/8/16): 8 is
    // 5-10 % ahead of 4 on the long shapes (book FFN1, vocabulary projection), level on the d=768 ones; 16 overflows the L2 on the book shapes.
    const int GN = g.gn > 0 ? g.gn : 8;
    const int grp = id / (GN * tilesM);
    const int gw = min(GN, tilesN - grp * GN);
    const int local = id - grp * GN * tilesM;
    const int tm = local / gw, tn = grp * GN + local % gw;
    const int m0 = tm * BM, n0 = tn * BN;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // provably wave-uniform (LDS DMA base)
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;

    // per-lane source rows / slots of this wave's chunks (constant over k): chunk c covers tile rows
    // 8c..8c+7; lane l lands at (row 8c + l/8, physical slot l%8) and therefore fetches logical slot
    // (l%8) ^ (row & 7).  Rows past the matrix edge re-read the last valid row (never stored).
    const int lrow = lane >> 3, lps = lane & 7;
    const bf16_t* asrc[A_CH];
    const bf16_t* bsrc[B_CH];
    if constexpr (!TT) {
#pragma unroll
        for (int c = 0; c < A_CH; ++c) {
            const int r = (wave * A_CH + c) * 8 + lrow;
            asrc[c] = (bz ? g.A2 : g.A) + (long)min(m0 + r, g.M - 1) * g.lda + ((lps ^ (r & 7)) << 3);
        }
        if constexpr (BF8) {
            // 16-row pieces of 64-byte rows: lane l lands at (row 16 c + l/4, physical 16-byte slot l%4) and fetches logical slot
            // (l%4) ^ g(l >> 4) of that row; bsrc counts BYTES here (bf16_t* arithmetic below is done on a byte pointer)
            const int gsw = (0x1230 >> (4 * (lane >> 4))) & 3;
#pragma unroll
            for (int c = 0; c < B_CH; ++c) {
                const int r = (wave * B_CH + c) * 16 + (lane >> 2);
                const unsigned char* bq = reinterpret_cast<const unsigned char*>(g.B);
                bsrc[c] = reinterpret_cast<const bf16_t*>(bq + (long)min(n0 + r, g.N - 1) * g.ldb + (((lane & 3) ^ gsw) << 4));
            }
        } else {
#pragma unroll
        for (int c = 0; c < B_CH; ++c) {
            const int r = (wave * B_CH + c) * 8 + lrow;
            bsrc[c] = (bz ? g.B2 : g.B) + (long)min(n0 + r, g.N - 1) * g.ldb + ((lps ^ (r & 7)) << 3);
        }
        }
    } else {
        // chunk ci = 4 k-rows of 256 bytes; lane l lands at (row 4 ci + l/16, physical 16-byte chunk l%16) and
        // fetches logical chunk (l%16) ^ f(row), f(row) = ((row & 3) << 2) | ((row >> 2) & 3).  Column chunks
        // past the matrix edge re-read the last whole chunk (their products are never stored).
        const int tr = lane >> 4, pc = lane & 15;
#pragma unroll
        for (int c = 0; c < A_CH; ++c) {
            const int r = (wave * A_CH + c) * 4 + tr;
            const int lc = pc ^ (((r & 3) << 2) | ((r >> 2) & 3));
            asrc[c] = g.A + (long)r * g.lda + min(m0 + lc * 8, g.M - 8);
        }
#pragma unroll
        for (int c = 0; c < B_CH; ++c) {
            const int r = (wave * B_CH + c) * 4 + tr;
            const int lc = pc ^ (((r & 3) << 2) | ((r >> 2) & 3));
            bsrc[c] = g.B + (long)r * g.ldb + min(n0 + lc * 8, g.N - 8);
        }
    }
    const int kbeg = by * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nk = (kend - kbeg) / BBK;

    auto issue = [&](int t) {
        if (BGEMM_ABL(g) & 1) return;
        char* st = smem + (t % NSTAGE) * ST_BYTES;
        const int k = kbeg + t * BBK;
        const long ka = TT ? (long)k * g.lda : k, kb = TT ? (long)k * g.ldb : k;     // TT: k advances rows
#pragma unroll
        for (int c = 0; c < A_CH; ++c)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(asrc[c] + ka), (lds_ptr_t)(st + (wave * A_CH + c) * 1024), 16, 0, 0);
#pragma unroll
        for (int c = 0; c < B_CH; ++c) {
            if constexpr (BF8)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(reinterpret_cast<const unsigned char*>(bsrc[c]) + kb), (lds_ptr_t)(st + A_BYTES + (wave * B_CH + c) * 1024), 16, 0, 0);
            else
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bsrc[c] + kb), (lds_ptr_t)(st + A_BYTES + (wave * B_CH + c) * 1024), 16, 0, 0);
        }
    };

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    issue(0);
#pragma unroll
    for (int pt = 1; pt < NSTAGE - 1; ++pt)
        if (pt < nk) issue(pt);
    for (int t = 0; t < nk; ++t) {
        // tile t has landed once all but the (one) newer tile's loads of THIS wave are done, and every
        // wave has said so at the barrier; the barrier also retires all reads of tile t-1, whose ring
        // slot tile t+2 may now overwrite.
        // tiles t+1 .. t+NSTAGE-2 (those that exist) are newer than tile t and may stay in flight
        {
            const int newer = min(NSTAGE - 2, nk - 1 - t);
            if (NSTAGE > 3 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LOADS) : "memory");
            else if (NSTAGE > 2 && newer >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        if (t + NSTAGE - 1 < nk) issue(t + NSTAGE - 1);
        const unsigned As = lds_base + (t % NSTAGE) * ST_BYTES;
        const unsigned Bs = As + A_BYTES;
        u32x4_t af[2][TM], bfr[2][TN];
        u32x2_t bq8[2][BF8 ? TN : 1];
        auto widen = [&](int kk) {                        // fp8 x 8 -> bf16 x 8, after the reads of half kk have landed
            if constexpr (BF8) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const auto lo0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)bq8[kk][j].x, false), lo1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)bq8[kk][j].x, true);
                    const auto hi0 = __builtin_amdgcn_cvt_pk_f32_fp8((int)bq8[kk][j].y, false), hi1 = __builtin_amdgcn_cvt_pk_f32_fp8((int)bq8[kk][j].y, true);
                    u32x4_t w;
                    w.x = (uint32_t)f2bf16(lo0[0]) | ((uint32_t)f2bf16(lo0[1]) << 16);
                    w.y = (uint32_t)f2bf16(lo1[0]) | ((uint32_t)f2bf16(lo1[1]) << 16);
                    w.z = (uint32_t)f2bf16(hi0[0]) | ((uint32_t)f2bf16(hi0[1]) << 16);
                    w.w = (uint32_t)f2bf16(hi1[0]) | ((uint32_t)f2bf16(hi1[1]) << 16);
                    bfr[kk][j] = w;
                }
            }
        };
        auto read_half = [&](int kk) {
            if constexpr (!TT) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[kk][i] = lds_read128(As + blds_off(wm * WMR + i * 16 + lr, kk * 4 + lq));
                if constexpr (BF8) {
                    // row r = wn * BN/2 + 16 j + lr: swizzle key (r >> 2) & 3 = (lr >> 2) & 3; this lane's 8 k-values are bytes
                    // 32 kk + 8 lq .. + 7 of the row = 16-byte slot 2 kk + (lq >> 1), half lq & 1
                    const int fsw = (0x1230 >> (4 * ((lr >> 2) & 3))) & 3;
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const unsigned ad = Bs + (wn * (BN / 2) + j * 16 + lr) * 64 + (((kk * 2 + (lq >> 1)) ^ fsw) << 4) + ((lq & 1) << 3);
                        u32x2_t q;
                        asm volatile("ds_read_b64 %0, %1" : "=v"(q) : "v"(ad) : "memory");
                        bq8[kk][j] = q;
                    }
                } else {
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[kk][j] = lds_read128(Bs + blds_off(wn * (BN / 2) + j * 16 + lr, kk * 4 + lq));
                }
            } else {
                // 16-lane group lq owns k = 32 kk + 8 lq .. + 7: two transposed 4-row blocks; lane 4q+p of the group
                // supplies the address of block row q, columns 4p .. 4p+3 and receives column (lane % 16)
                const int q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int c0 = (wm * (BM / 2) + i * 16) / 8 + (p4 >> 1);
                    const u32x2_t lo = lds_read64_tr(As + tlds_off(kk * 32 + lq * 8 + q4, c0) + 8 * (p4 & 1));
                    const u32x2_t hi = lds_read64_tr(As + tlds_off(kk * 32 + lq * 8 + 4 + q4, c0) + 8 * (p4 & 1));
                    af[kk][i] = (u32x4_t){lo.x, lo.y, hi.x, hi.y};
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int c0 = (wn * (BN / 2) + j * 16) / 8 + (p4 >> 1);
                    const u32x2_t lo = lds_read64_tr(Bs + tlds_off(kk * 32 + lq * 8 + q4, c0) + 8 * (p4 & 1));
                    const u32x2_t hi = lds_read64_tr(Bs + tlds_off(kk * 32 + lq * 8 + 4 + q4, c0) + 8 * (p4 & 1));
                    bfr[kk][j] = (u32x4_t){lo.x, lo.y, hi.x, hi.y};
                }
            }
        };
        // NT: both 32-wide k halves are requested before the first MFMA (the second half's reads retire under
        // the first half's MFMAs); TT issues twice as many (64-bit) reads, more than lgkmcnt can count: half by half
        if (!(BGEMM_ABL(g) & 4)) read_half(0);
        if constexpr (!TT) {
            if (!(BGEMM_ABL(g) & 4)) read_half(1);
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN) : "memory");      // first half landed
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        widen(0);
        if (!(BGEMM_ABL(g) & 2))
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[0][i]),
                                                                    __builtin_bit_cast(bf16x8_t, bfr[0][j]), acc[i][j], 0, 0, 0);
        if constexpr (TT) read_half(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        widen(1);
        if (!(BGEMM_ABL(g) & 2))
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[1][i]),
                                                                    __builtin_bit_cast(bf16x8_t, bfr[1][j]), acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                  // all tile reads done before smem is reused for C
    if (BGEMM_ABL(g) & 8) return;

    if (g.splits > 1 || g.slab_only) {
        float* slab = g.slab + (((long)bz * g.splits + by) * g.M) * g.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 16 + lr;
                if (n >= g.N) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * WMR + i * 16 + lq * 4 + r;
                    if (m < g.M) {
                        if constexpr (SLAB_THROUGH) __hip_atomic_store(&slab[(long)m * g.N + n], acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else slab[(long)m * g.N + n] = acc[i][j][r];
                    }
                }
            }
        return;
    }

    // epilogue through LDS: 256-byte fp32 / 128-byte bf16 row segments per store instruction
    const uint32_t dseed = g.drop.p > 0.f ? cst_drop_seed(g.drop) : 0u;
    constexpr int WM = WMR, WN = BN / 2, CLD = WN;
    static_assert(NW * WM * CLD * 4 <= NSTAGE * ST_BYTES, "C staging must fit the ring");
    float* Cs = reinterpret_cast<float*>(smem) + wave * WM * CLD;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cs[(i * 16 + lq * 4 + r) * CLD + j * 16 + lr] = acc[i][j][r];
    __builtin_amdgcn_wave_barrier();
    constexpr int C4 = WN / 4;
    const bool vec = bgemm_vec_ok(g);
    static_assert(CLD == 64, "bgemm_write_rows assumes 256-byte staging rows");
    if (vec && m0 + wm * WM + WM <= g.M && n0 + wn * WN + WN <= g.N) {        // wave-uniform: the whole sub-tile is inside the matrix
        float s4[4], b4[4];
        bgemm_col_operands(g, n0 + wn * WN, lane, s4, b4);
        bgemm_write_rows<WM>(g, dseed, lds_base + wave * WM * CLD * 4, m0 + wm * WM, n0 + wn * WN, lane, s4, b4);
        return;
    }
#pragma unroll
    for (int it = 0; it < WM * C4 / 64; ++it) {
        const int idx = lane + 64 * it;
        const int rr = idx / C4, cc = (idx % C4) * 4;
        const int m = m0 + wm * WM + rr, n = n0 + wn * WN + cc;
        if (m >= g.M || n >= g.N) continue;
        const float4 a4 = *reinterpret_cast<const float4*>(&Cs[rr * CLD + cc]);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
        if (vec && n + 3 < g.N) bgemm_store4(g, dseed, m, n, av);
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < g.N) bgemm_store(g, dseed, m, n + e, av[e]);
        }
    }
}

// consecutive workgroup ids go round the eight XCDs: id = the workgroup's place in an order that keeps each XCD's workgroups together
__device__ __forceinline__ int bgemm_xcd_order(const int bid, const int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

template <int BM, int BN, int NSTAGE, bool TT = false, bool BF8 = false, int NW = 4>
__global__ __launch_bounds__(64 * NW) void cst_gemm_bf16_kernel(BGemmArgs g) {
    bgemm_tile_body<BM, BN, NSTAGE, TT, BF8, NW>(g, bgemm_xcd_order(blockIdx.x, gridDim.x), blockIdx.y, blockIdx.z);
}

// Several weight-gradient products C_p = A_p^T B_p in ONE launch (cst_gemm_bf16_tt_group_*): the four dW of an encoder layer each have
// 36-108 output tiles of 128 x 128 over a long contraction (the token count) -- launched one by one each needs split-K slabs and a
// reduce launch to fill the CUs; together they are 336 tiles, enough for whole-K workgroups (measured, tools/tt_group_probe.py:
// 193.6 -> 128.7 us for the Matcher's layer, 119.5 -> 68.9 us for the MLM's).  The workgroups of the launch are numbered through the
// problems in order; each finds its problem from the prefix sums of the tile counts and runs the one-product body on it.
#define CST_TT_GROUP_MAX 8
struct BTtProblem { const bf16_t* A; const bf16_t* B; float* C; long lda, ldb, ldc; int M, N, K, accumulate; float* slab; int kps; int cvec; };
struct BTtGroup { BTtProblem p[CST_TT_GROUP_MAX]; int start[CST_TT_GROUP_MAX + 1]; int n; int splits; int* counters; };

// splits == 1: one workgroup per output tile, the whole contraction.  splits = S > 1 (fewer tiles than fill the CUs twice over): S
// workgroups per tile, each writes its partial tile to the problem's slab, and the one that finishes LAST (a counter per tile, no
// waiting) adds the S partials in split order and writes C -- the sum does not depend on which workgroup that is.  The counter is
// left at zero for the next launch.
__global__ __launch_bounds__(256) void cst_gemm_bf16_tt_group_kernel(BTtGroup q) {
    const int id = bgemm_xcd_order(blockIdx.x, gridDim.x);
    int p = 0;
#pragma unroll
    for (int i = 1; i < CST_TT_GROUP_MAX; ++i)
        if (i < q.n && id >= q.start[i]) p = i;
    const BTtProblem& t = q.p[p];
    const int S = q.splits;
    const int local = id - q.start[p];
    const int tile = S > 1 ? local / S : local, by = local - tile * S;
    BGemmArgs g{};                                   // everything the weight gradients do not use is a compile-time null / zero here
    g.A = t.A; g.B = t.B; g.C = t.C; g.lda = t.lda; g.ldb = t.ldb; g.ldc = t.ldc;
    g.M = t.M; g.N = t.N; g.K = t.K; g.accumulate = t.accumulate;
    g.alpha = 1.f; g.gate_scale = 1.f; g.splits = S; g.k_per_split = t.kps; g.slab = t.slab;
    if (S == 1) {
        bgemm_tile_body<128, 128, 2, true, false, 4>(g, tile, 0, 0);
        return;
    }
    bgemm_tile_body<128, 128, 2, true, false, 4, true>(g, tile, by, 0);
    __shared__ int last_s;
    // the partial tile was written with device-scope stores: once they have completed (no cache to write back: a release fence here would
    // flush the XCD's whole L2, 0.2 us per workgroup -- measured) ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                 // ... for every thread of the workgroup, the workgroup is counted
    int* ctr = q.counters + (q.start[p] / S + tile);
    if (threadIdx.x == 0) {
        const int seen = atomicAdd(ctr, 1);
        last_s = seen == S - 1;
        if (seen == S - 1) *ctr = 0;                 // nobody else touches the counter any more in this launch
    }
    __syncthreads();
    if (!last_s) return;
    // the tile's origin, as bgemm_tile_body finds it
    const int tilesM = (t.M + 127) / 128, tilesN = (t.N + 127) / 128;
    const int GN = 8, grp = tile / (GN * tilesM), gw = min(GN, tilesN - grp * GN), loc = tile - grp * GN * tilesM;
    const int m0 = (loc / gw) * 128, n0 = (grp * GN + loc % gw) * 128;
    const long MN = (long)t.M * t.N;
#pragma unroll 4
    for (int e = threadIdx.x; e < 128 * 32; e += 256) {          // 128 rows x 32 column quads
        const int m = m0 + (e >> 5), n = n0 + ((e & 31) << 2);
        if (m >= t.M || n >= t.N) continue;                       // (N is a multiple of 8: a quad is inside or outside as a whole)
        const float* sp = t.slab + (long)m * t.N + n;
        auto ld4 = [](const float* a) {               // device-scope loads: from memory, never a stale line of this XCD's L2
            return make_float4(__hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(a + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                               __hip_atomic_load(a + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(a + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        };
        float4 v = ld4(sp);
        for (int k = 1; k < S; ++k) {
            const float4 w = ld4(sp + k * MN);
            v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
        }
        float* cp = t.C + (long)m * t.ldc + n;
        if (t.cvec) {
            if (t.accumulate) { const float4 o = *reinterpret_cast<const float4*>(cp); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *reinterpret_cast<float4*>(cp) = v;
        } else {
            if (t.accumulate) { v.x += cp[0]; v.y += cp[1]; v.z += cp[2]; v.w += cp[3]; }
            cp[0] = v.x; cp[1] = v.y; cp[2] = v.z; cp[3] = v.w;
        }
    }
}

// BATCH: four slab loads in flight at a time (same summation order).  A runtime-bounded loop of one load and one dependent
// add serialises the L2 latency of every split: worth it from 4 splits up (the long-K products run 16-24).
template <bool BATCH>
__global__ __launch_bounds__(256) void cst_gemm_bf16_reduce(BGemmArgs g) {
    const uint32_t dseed = g.drop.p > 0.f ? cst_drop_seed(g.drop) : 0u;
    const long MN = (long)g.M * g.N;
    if ((g.N & 3) == 0 && bgemm_vec_ok(g) && (reinterpret_cast<uintptr_t>(g.slab) & 15) == 0) {
        const long Q = MN >> 2;
        for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < Q; q += (long)gridDim.x * 256) {
            float4 acc = *reinterpret_cast<const float4*>(g.slab + 4 * q);
            int s = 1;
            if constexpr (BATCH) {
                const float* p = g.slab + 4 * q;
                for (; s + 3 < g.splits; s += 4) {
                    const float4 t0 = *reinterpret_cast<const float4*>(p + s * MN), t1 = *reinterpret_cast<const float4*>(p + (s + 1) * MN);
                    const float4 t2 = *reinterpret_cast<const float4*>(p + (s + 2) * MN), t3 = *reinterpret_cast<const float4*>(p + (s + 3) * MN);
                    acc.x = (((acc.x + t0.x) + t1.x) + t2.x) + t3.x; acc.y = (((acc.y + t0.y) + t1.y) + t2.y) + t3.y;
                    acc.z = (((acc.z + t0.z) + t1.z) + t2.z) + t3.z; acc.w = (((acc.w + t0.w) + t1.w) + t2.w) + t3.w;
                }
            }
            for (; s < g.splits; ++s) {
                const float4 t = *reinterpret_cast<const float4*>(g.slab + s * MN + 4 * q);
                acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
            }
            const long e = 4 * q;
            const float av[4] = {acc.x, acc.y, acc.z, acc.w};
            bgemm_store4(g, dseed, (int)(e / g.N), (int)(e % g.N), av);
        }
        return;
    }
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < MN; e += (long)gridDim.x * 256) {
        float acc = 0.f;
        for (int s = 0; s < g.splits; ++s) acc += g.slab[s * MN + e];
        bgemm_store(g, dseed, (int)(e / g.N), (int)(e % g.N), acc);
    }
}

// kernel-precise timing shared with gemm.hip (cst_gemm_profile_enable / _read)
#include <hip/hip_ext.h>

#ifdef CST_BENCH_VARIANTS
// =============================================================================================
// 256 x 256 tile, 8 waves, K-tiles of 32 in a 4-stage ring (after cdna_hip_programming.md section 5, "The 256^2 8-phase
// template"): the encoder-layer products with N >= 1536 (packed QKV, FFN1, and the dgrad of FFN2).
//
// Why: measured on these shapes (tools/gemm_bench.py enc) every tile of the small-tile kernel above, and a first 256^2
// version with two 64-deep K-tile buffers, sits at 20-40 GB/s of operand bytes per CU: the CUs of an XCD walk K in
// lockstep, so each K-tile is a first touch served at Infinity-Cache / HBM latency (~1.5-2.5 us under load), and the rate
// is (bytes in flight) / latency.  What can be raised is (a) FLOP per operand byte -- 128 at 256 x 256 against 43-64 at
// 64 x 128 / 128 x 128 -- and (b) bytes in flight per CU: four 32 KiB stages keep up to ~80 KiB in flight (a 64-deep
// double buffer caps it at 48), which is what the 32-deep K-tile buys.  A wave owns 128 x 64 of the tile: 32 MFMAs per
// K-tile against 12 fragment reads.
//
//   LDS: 4 stages x (A [256 rows][64 B] + B [256][64 B]) = 128 KiB.  A DMA piece (1 KiB) is 16 rows of 64 bytes; the four
//   16-byte slots of a row are XOR-ed with g((row >> 2) & 3), g = {0, 3, 2, 1}, which makes every ds_read_b128 lane group
//   ({0-3, 12-15, 20-27}, ... -- MI355X_MICROARCH.md LDS table) hit 16 distinct positions of the 256-byte bank line.
//   K-tile t, two phases of 16 MFMAs (row halves mh = 0, 1 of the wave tile), ONE barrier:
//     P0(t): DMA B(t+3) ; read A(mh 1, t) into the other A set               ; 16 MFMA on A(mh 0) x B
//     P1(t): vmcnt(6) -> tile t+1 landed ; barrier ; DMA A(t+3) ;
//            read A(mh 0, t+1) and B(t+1) into the other sets                 ; 16 MFMA on A(mh 1) x B
//   Stage (t+3) % 4 held tile t-1, whose last fragment read completed before the barrier of P1(t-1): both DMA issues are
//   WAR-safe without a barrier of their own.  Never vmcnt(0) inside the loop.
// Requires M % 256 == 0, N % 256 == 0, K % 32 == 0 (K % 64 == 0 as everywhere), no split-K.
// =============================================================================================
constexpr int GB_T = 256;                 // tile edge
constexpr int GB_K = 32;                  // K-tile
constexpr int GB_ROW = 64;                // bytes per LDS row
constexpr int GB_ABYTES = GB_T * GB_ROW;  // 16 KiB per operand per K-tile
constexpr int GB_STAGE = 2 * GB_ABYTES;   // 32 KiB per K-tile
constexpr int GB_NST = 4;

template <int OFF>
__device__ __forceinline__ u32x4_t lds_read128_off(unsigned addr) {
    u32x4_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}

struct GbFrag {
    u32x4_t a[2][4];       // [set][row tile]
    u32x4_t b[2][4];       // [set][column tile]
};

template <int SET, int MH>
__device__ __forceinline__ void gb_read_a(GbFrag& f, unsigned ad) {
    f.a[SET][0] = lds_read128_off<(MH * 64 + 0) * GB_ROW>(ad);
    f.a[SET][1] = lds_read128_off<(MH * 64 + 16) * GB_ROW>(ad);
    f.a[SET][2] = lds_read128_off<(MH * 64 + 32) * GB_ROW>(ad);
    f.a[SET][3] = lds_read128_off<(MH * 64 + 48) * GB_ROW>(ad);
}
template <int SET>
__device__ __forceinline__ void gb_read_b(GbFrag& f, unsigned ad) {
    f.b[SET][0] = lds_read128_off<0 * GB_ROW>(ad);
    f.b[SET][1] = lds_read128_off<16 * GB_ROW>(ad);
    f.b[SET][2] = lds_read128_off<32 * GB_ROW>(ad);
    f.b[SET][3] = lds_read128_off<48 * GB_ROW>(ad);
}
template <int SA, int SB, int MH>
__device__ __forceinline__ void gb_mfma(const GbFrag& f, f32x4_t (&acc)[8][4]) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[MH * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, f.a[SA][i]),
                                                                         __builtin_bit_cast(bf16x8_t, f.b[SB][j]), acc[MH * 4 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

// WR = wave rows: 2 -> 256 x 256 tile, 8 waves, one workgroup per CU (NST = 4 stages of 32 KiB);
//                 1 -> 128 x 256 tile, 4 waves, TWO workgroups per CU (NST = 3 stages of 24 KiB): the two run out of phase, so one's
//                      prologue / C write-out (32-75 MB of fp32 per launch, the largest HBM stream of these products) overlaps the
//                      other's K-loop -- at 256 x 256 every CU computes, then every CU writes.
template <int WR, int NST>
__global__ __launch_bounds__(256 * WR, 2) void cst_gemm_bf16_big_kernel(BGemmArgs g, int gn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int BM = 128 * WR, NW = 4 * WR;
    constexpr int A_BYTES = BM * GB_ROW, B_BYTES = GB_T * GB_ROW, STAGE = A_BYTES + B_BYTES;
    constexpr int PA = BM / 16 / NW, PB = GB_T / 16 / NW;      // DMA pieces (16 rows) per wave per K-tile
    constexpr int D = NST - 1;                                  // prefetch distance in K-tiles
    static_assert(D == 2 || D == 3, "wait counts below are written out for prefetch distances 2 and 3");
    const int abl = gn >> 16;               // timing ablations (CST_GB_ABL): 1 = no DMA, 2 = no MFMA, 4 = no fragment reads
    gn &= 0xffff;
    const int tilesM = g.M / BM, tilesN = g.N / GB_T;
    int id;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    // gn tile columns per group: an XCD (consecutive ids) works on a gn-wide strip, sharing few A and B panels
    const int grp = id / (gn * tilesM);
    const int gw = min(gn, tilesN - grp * gn);
    const int local = id - grp * gn * tilesM;
    const int tm = local / gw, tn = grp * gn + local % gw;
    const int m0 = tm * BM, n0 = tn * GB_T;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lr = lane & 15, lq = lane >> 4;

    // DMA sources: lane l of a piece lands at (row r0 + l/4, physical slot l%4); r0 is a multiple of 16, so the row's swizzle
    // key ((r0 + l/4) >> 2) & 3 is l >> 4 and the lane fetches logical slot (l%4) ^ g(l >> 4): one per-lane base per operand.
    const int gsw = (0x1230 >> (4 * (lane >> 4))) & 3;                 // g = {0, 3, 2, 1}
    const int src_slot = (lane & 3) ^ gsw;
    const bf16_t* abase = g.A + (long)(m0 + (lane >> 2)) * g.lda + (src_slot << 3);
    const bf16_t* bbase = g.B + (long)(n0 + (lane >> 2)) * g.ldb + (src_slot << 3);
    const int nk = g.K / GB_K;
    auto issue_a = [&](int t) {
        if (abl & 1) return;
        char* st = smem + (t % NST) * STAGE;
        const long k = (long)t * GB_K;
#pragma unroll
        for (int c = 0; c < PA; ++c) {
            const int r0 = (PA * wave + c) * 16;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(abase + (long)r0 * g.lda + k), (lds_ptr_t)(st + r0 * GB_ROW), 16, 0, 0);
        }
    };
    auto issue_b = [&](int t) {
        if (abl & 1) return;
        char* st = smem + (t % NST) * STAGE + A_BYTES;
        const long k = (long)t * GB_K;
#pragma unroll
        for (int c = 0; c < PB; ++c) {
            const int r0 = (PB * wave + c) * 16;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bbase + (long)r0 * g.ldb + k), (lds_ptr_t)(st + r0 * GB_ROW), 16, 0, 0);
        }
    };

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    GbFrag f;

    // fragment read addresses (stage 0): row (wr * 128 + lr) resp. (wc * 64 + lr), slot lq ^ g((lr >> 2) & 3); the row half and the
    // row / column tile go into the instruction's offset field (they are multiples of 16 rows: the swizzle key is unchanged)
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const int fsw = (0x1230 >> (4 * ((lr >> 2) & 3))) & 3;
    const unsigned a_ad = lds_base + (wr * 128 + lr) * GB_ROW + ((lq ^ fsw) << 4);
    const unsigned b_ad = lds_base + A_BYTES + (wc * 64 + lr) * GB_ROW + ((lq ^ fsw) << 4);

    // prologue: tiles 0 .. D-1 in flight; tile 0 must have landed
    issue_a(0); issue_b(0);
    if (nk > 1) { issue_b(1); issue_a(1); }
    if (D > 2 && nk > 2) { issue_b(2); issue_a(2); }
    {
        const int younger = min(nk, D) - 1;                // whole tiles issued after tile 0
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (PA + PB)) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    gb_read_a<0, 0>(f, a_ad);
    gb_read_b<0>(f, b_ad);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

#define GB_PHASE_END()                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);
    // P1's wait: tile t+1 has landed; younger pieces of this wave still in flight: the whole tiles t+2 .. t+D-1 and B(t+D)
#define GB_WAIT_NEXT()                                                                                          \
    if constexpr (D == 3) {                                                                                     \
        if (t + 3 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + 2 * PB) : "memory");                      \
        else if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PA + PB) : "memory");                     \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
    } else {                                                                                                    \
        if (t + 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PB) : "memory");                               \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                   \
    }
    // one K-tile; SB = the B register set holding tile t (tile t+1 goes to the other one)
#define GB_TILE(SB)                                                                                             \
    {                                                                                                           \
        const unsigned cur = (t % NST) * STAGE, nxt = ((t + 1) % NST) * STAGE;                                  \
        if (t + D < nk) issue_b(t + D);                                                                         \
        if (!(abl & 4)) gb_read_a<1, 1>(f, a_ad + cur);                                                         \
        if (!(abl & 2)) gb_mfma<0, SB, 0>(f, acc);                                                                            \
        GB_PHASE_END()                                                                                          \
        GB_WAIT_NEXT()                                                                                          \
        __builtin_amdgcn_s_barrier();                                                                           \
        if (t + D < nk) issue_a(t + D);                                                                         \
        if (t + 1 < nk && !(abl & 4)) {                                                                         \
            gb_read_a<0, 0>(f, a_ad + nxt);                                                                     \
            gb_read_b<1 - SB>(f, b_ad + nxt);                                                                   \
        }                                                                                                       \
        if (!(abl & 2)) gb_mfma<1, SB, 1>(f, acc);                                                                            \
        GB_PHASE_END()                                                                                          \
    }
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        GB_TILE(0)
        ++t;
        GB_TILE(1)
        --t;
    }
    if (t < nk) GB_TILE(0)
#undef GB_TILE
#undef GB_WAIT_NEXT
#undef GB_PHASE_END
    __syncthreads();                                  // every fragment read is done before the ring becomes C staging
    if (abl & 8) return;                              // ablation: no write-out

    // epilogue: the wave tile's two 64 x 64 halves through LDS (16 KiB per wave), whole 256-byte row segments per store
    const uint32_t dseed = g.drop.p > 0.f ? cst_drop_seed(g.drop) : 0u;
    constexpr int CLD = 64, C4 = 16;
    static_assert(NW * 64 * CLD * 4 <= NST * STAGE, "C staging must fit the ring");
    float* Cs = reinterpret_cast<float*>(smem) + wave * 64 * CLD;
    const bool vec = bgemm_vec_ok(g);
    float s4[4], b4[4];
    if (vec) bgemm_col_operands(g, n0 + wc * 64, lane, s4, b4);
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) Cs[(i * 16 + lq * 4 + r) * CLD + j * 16 + lr] = acc[mh * 4 + i][j][r];
        __builtin_amdgcn_wave_barrier();
        if (vec) bgemm_write_rows<64>(g, dseed, lds_base + wave * 64 * CLD * 4, m0 + wr * 128 + mh * 64, n0 + wc * 64, lane, s4, b4);
        else {
#pragma unroll 4
            for (int it = 0; it < 64 * C4 / 64; ++it) {
                const int idx = lane + 64 * it;
                const int rr = idx / C4, cc = (idx % C4) * 4;
                const int m = m0 + wr * 128 + mh * 64 + rr, n = n0 + wc * 64 + cc;
                const float4 a4 = *reinterpret_cast<const float4*>(&Cs[rr * CLD + cc]);
#pragma unroll
                for (int e = 0; e < 4; ++e) bgemm_store(g, dseed, m, n + e, (&a4.x)[e]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// =============================================================================================
// Loader / consumer, persistent: 256 x 128 tiles, 4 MFMA waves + 4 DMA waves per workgroup, one workgroup per CU walking
// its share of the tiles (tile code 248).
//
// Why (profiles/round2_gemm_ablation.txt): in the kernels above a product costs (L2 -> LDS fill) + (C write-out) -- the MFMAs
// are hidden, but the workgroups of a CU fill in lockstep and then write in lockstep, and a wave cannot do both at once:
// gfx950 counts loads and stores in one vmcnt, so a wave with stores in flight cannot wait for a younger load without waiting
// for the stores.  Here the roles are split: waves 4-7 only issue LDS-DMA and count vmcnt (loads only), waves 0-3 only read
// fragments, issue MFMAs and write C (stores only, never waited for).  The K-tile stream runs on across tile boundaries, so
// while the MFMA waves write a finished tile out the DMA waves already fill the next tile's first K-tiles.
//   K-tile = 64 (full 128-byte rows: half-line rows, as the 256-wide kernels above use, fetch every line twice and were measured
//   at half the fill rate).  LDS: 3 stages x (A [256][128 B] + B [128][128 B]) = 144 KiB, image and swizzle of the small-tile
//   kernel (blds_off).  The stage a tile's last K-tile was read from doubles as its C staging (the loaders get it back at the
//   next barrier, which the MFMA waves only reach after the write-out).
//   MFMA waves, K-tile g: barrier(g) ; read A(row half 0) and B of k-half 0 ; then per 32-deep half: read A(row half 1) ; 16 MFMA (row
//                      half 0) ; read A(row half 0) and B of the next half ; 16 MFMA (row half 1).  Nothing of K-tile g + 1 is read before
//                      barrier(g + 1): one exposed fragment read per K-tile buys a second K-tile in flight (three stages: one being
//                      read, two landing).
//   DMA waves, step g: vmcnt(pieces of K-tile g + 1) -> K-tile g landed ; barrier(g) ; DMA K-tile g + 2 into the stage K-tile g - 1 left
// Requires M % 256 == 0, N % 128 == 0, K % 64 == 0, no split-K.
// =============================================================================================
__device__ __forceinline__ void lds_write32(unsigned addr, float v) {
    asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

constexpr int LC_NST = 3;
constexpr int LC_BM = 256, LC_BN = 128;
constexpr int LC_A_BYTES = LC_BM * BROW, LC_B_BYTES = LC_BN * BROW, LC_STAGE = LC_A_BYTES + LC_B_BYTES;     // 32 + 16 KiB

__global__ __launch_bounds__(512) void cst_gemm_bf16_lc_kernel(BGemmArgs g, int gn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = LC_NST, D = NST - 1, BM = LC_BM, BN = LC_BN, STAGE = LC_STAGE;
    constexpr int PA = BM / 8 / 4, PB = BN / 8 / 4, PT = PA + PB;            // 1-KiB pieces (8 rows) per loader wave per K-tile: 8 + 4
    static_assert(D == 2, "the vmcnt cases below are written for a prefetch distance of 2");
    const int tilesM = g.M / BM, tilesN = g.N / BN, ntiles = tilesM * tilesN;
    const int nk = g.K / BBK;
    int vb;                                                // virtual workgroup id: consecutive ids share an XCD
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        vb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    const int mytiles = (ntiles - vb + (int)gridDim.x - 1) / (int)gridDim.x;
    const int G = mytiles * nk;                            // K-tiles this workgroup streams
    auto tile_origin = [&](int j, int& m0, int& n0) {      // j-th tile of this workgroup
        const int id = vb + j * (int)gridDim.x;
        const int grp = id / (gn * tilesM);
        const int gw = min(gn, tilesN - grp * gn);
        const int local = id - grp * gn * tilesM;
        m0 = (local / gw) * BM;
        n0 = (grp * gn + local % gw) * BN;
    };
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (G == 0) return;

    if (wave >= 4) {
        // ------------------------------------------------------------------------------ loaders
        const int lw = wave - 4;
        const int lrow = lane >> 3, lps = lane & 7;        // lane lands at (row 8 p + lrow, physical slot lps): fetches logical slot lps ^ lrow
        const bf16_t* abase = nullptr;
        const bf16_t* bbase = nullptr;
        int cur_j = -1;
        auto issue = [&](int gk) {
            const int j = gk / nk, t = gk - j * nk;
            if (j != cur_j) {
                int m0, n0;
                tile_origin(j, m0, n0);
                abase = g.A + (long)(m0 + lrow) * g.lda + ((lps ^ lrow) << 3);
                bbase = g.B + (long)(n0 + lrow) * g.ldb + ((lps ^ lrow) << 3);
                cur_j = j;
            }
            char* st = smem + (gk % NST) * STAGE;
            const long k = (long)t * BBK;
#pragma unroll
            for (int c = 0; c < PA; ++c) {
                const int r0 = (PA * lw + c) * 8;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(abase + (long)r0 * g.lda + k), (lds_ptr_t)(st + r0 * BROW), 16, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < PB; ++c) {
                const int r0 = (PB * lw + c) * 8;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(bbase + (long)r0 * g.ldb + k), (lds_ptr_t)(st + LC_A_BYTES + r0 * BROW), 16, 0, 0);
            }
        };
        auto wait_younger = [&](int younger) {             // all but the `younger` most recent K-tiles of this wave have landed
            if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PT) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        };
        const int pre = min(D, G);
        for (int gk = 0; gk < pre; ++gk) issue(gk);
        for (int gk = 0; gk < G; ++gk) {
            // issued so far: K-tiles 0 .. min(gk + D, G) - 1; K-tile gk must have landed -- the one issued after it may stay in flight
            wait_younger(min(gk + D, G) - 1 - gk);
            __builtin_amdgcn_s_barrier();                  // barrier(gk): K-tile gk is in LDS; the MFMA waves are done with K-tile gk - 1
            if (gk + D < G) issue(gk + D);                 // into the stage K-tile gk - 1 left
        }
        return;
    }

    // ---------------------------------------------------------------------------------- consumers
    const int wr = wave >> 1, wc = wave & 1;               // wave tile: rows wr * 128 .. + 127, columns wc * 64 .. + 63
    const int lr = lane & 15, lq = lane >> 4;
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    // fragment of row (16 t + lr), k-half kk: 16-byte slot (4 kk + lq) ^ (lr & 7)
    const unsigned a_ad = lds_base + (wr * 128 + lr) * BROW + ((lq ^ (lr & 7)) << 4);
    const unsigned b_ad = lds_base + LC_A_BYTES + (wc * 64 + lr) * BROW + ((lq ^ (lr & 7)) << 4);
    const unsigned a_ad1 = lds_base + (wr * 128 + lr) * BROW + (((4 + lq) ^ (lr & 7)) << 4);               // k-half 1
    const unsigned b_ad1 = lds_base + LC_A_BYTES + (wc * 64 + lr) * BROW + (((4 + lq) ^ (lr & 7)) << 4);
    const uint32_t dseed = g.drop.p > 0.f ? cst_drop_seed(g.drop) : 0u;

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    struct { u32x4_t a[2][4]; u32x4_t b[2][4]; } f;        // a[row half][row tile], b[k-half parity][column tile]

#define LC_READ_A(MH, AD)                                                                          \
    {                                                                                              \
        const unsigned aa = (AD);                                                                  \
        f.a[MH][0] = lds_read128_off<(MH * 64 + 0) * BROW>(aa);   f.a[MH][1] = lds_read128_off<(MH * 64 + 16) * BROW>(aa);  \
        f.a[MH][2] = lds_read128_off<(MH * 64 + 32) * BROW>(aa);  f.a[MH][3] = lds_read128_off<(MH * 64 + 48) * BROW>(aa);  \
    }
#define LC_READ_B(SET, AD)                                                                         \
    {                                                                                              \
        const unsigned bb = (AD);                                                                  \
        f.b[SET][0] = lds_read128_off<0 * BROW>(bb);      f.b[SET][1] = lds_read128_off<16 * BROW>(bb);     \
        f.b[SET][2] = lds_read128_off<32 * BROW>(bb);     f.b[SET][3] = lds_read128_off<48 * BROW>(bb);     \
    }
#define LC_MFMA(MH, SB)                                                                            \
    __builtin_amdgcn_s_setprio(1);                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                  \
        acc[MH * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, f.a[MH][i]), \
                                                                      __builtin_bit_cast(bf16x8_t, f.b[SB][j]), acc[MH * 4 + i][j], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);
#define LC_WAIT_LDS()                                                                              \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                             \
    __builtin_amdgcn_sched_barrier(0);
    // write-out of tile j through the stage K-tile gk was read from (8 KiB per wave, 32-row quarters of the 128 x 64 wave tile)
    auto epilogue = [&](int j, int gk) {
        int m0, n0;
        tile_origin(j, m0, n0);
        const unsigned cs = lds_base + (gk % NST) * STAGE + wave * 32 * 64 * 4;
        const unsigned cw = cs + (lq * 4 * 64 + lr) * 4;
        float s4[4], b4[4];
        bgemm_col_operands(g, n0 + wc * 64, lane, s4, b4);      // the one vmcnt wait of the write-out (the previous tile's stores are long done)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int r = 0; r < 4; ++r) lds_write32(cw + ((i * 16 + r) * 64 + jj * 16) * 4, acc[q * 2 + i][jj][r]);
            bgemm_write_rows_t<32, false>(g, dseed, cs, m0 + wr * 128 + q * 32, n0 + wc * 64, lane, s4, b4);   // launcher: bgemm_vec_ok, no per-element operands
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) acc[i][jj] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    };
    int kt = 0, jt = 0;
    for (int gk = 0; gk < G; ++gk) {
        const unsigned so = (gk % NST) * STAGE;
        __builtin_amdgcn_s_barrier();                      // barrier(gk): K-tile gk has landed
        LC_READ_A(0, a_ad + so)
        LC_READ_B(0, b_ad + so)
        LC_WAIT_LDS()
        // ---- k-half 0 (B set 0)
        LC_READ_A(1, a_ad + so)
        LC_MFMA(0, 0)
        LC_WAIT_LDS()
        LC_READ_A(0, a_ad1 + so)
        LC_READ_B(1, b_ad1 + so)
        LC_MFMA(1, 0)
        LC_WAIT_LDS()
        // ---- k-half 1 (B set 1)
        LC_READ_A(1, a_ad1 + so)
        LC_MFMA(0, 1)
        LC_WAIT_LDS()
        LC_MFMA(1, 1)
        if (++kt == nk) { epilogue(jt, gk); kt = 0; ++jt; }
    }
#undef LC_MFMA
#undef LC_READ_A
#undef LC_READ_B
#undef LC_WAIT_LDS
}

static bool bgemm_lc_ok(const BGemmArgs& g) {
    return g.M % LC_BM == 0 && g.N % LC_BN == 0 && g.K % BBK == 0 && g.splits == 1 && !g.slab_only && !g.A2 && bgemm_vec_ok(g) &&
           !g.addend && g.act < 3 && !(g.C && g.accumulate);       // the MFMA waves' write-out holds no global load
}

static int bgemm_lc_launch(const BGemmArgs& g, hipStream_t st) {
    const size_t lds = (size_t)LC_NST * LC_STAGE;
    static const int gn = getenv("CST_GB_GN") ? atoi(getenv("CST_GB_GN")) : 2;
    static const int ncu = []() { int dev = 0, n = 256; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
    static CstPerDevice attr_done;
    if (cst_first_on_device(attr_done)) {
        (void)hipFuncSetAttribute((const void*)cst_gemm_bf16_lc_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    const int ntiles = (g.M / LC_BM) * (g.N / LC_BN);
    dim3 grid(ntiles < ncu ? ntiles : ncu, 1, 1), block(512);
    if (cst_prof_on()) {
        hipEvent_t ea, eb;
        (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
        cst_prof_push_shape(ea, eb, 2.0 * g.M * g.N * g.K, 2.0 * ((double)g.M * g.K + (double)g.N * g.K) + (g.C ? 4.0 : 0.0) * g.M * g.N + (g.Cb ? 2.0 : 0.0) * g.M * g.N, 1,
                            g.M, g.N, g.K);
        hipExtLaunchKernelGGL(cst_gemm_bf16_lc_kernel, grid, block, lds, st, ea, eb, 0, g, gn);
    } else {
        hipLaunchKernelGGL(cst_gemm_bf16_lc_kernel, grid, block, lds, st, g, gn);
    }
    return 0;
}

static bool bgemm_big_ok(const BGemmArgs& g, int wr) {
    return g.M % (128 * wr) == 0 && g.N % GB_T == 0 && g.K % GB_K == 0 && g.splits == 1 && !g.slab_only && !g.A2;
}

template <int WR, int NST>
static int bgemm_big_launch_t(const BGemmArgs& g, hipStream_t st) {
    const size_t lds = (size_t)NST * (128 * WR + GB_T) * GB_ROW;
    static const int gn0 = getenv("CST_GB_GN") ? atoi(getenv("CST_GB_GN")) : 2;       // tile columns per XCD strip (A/B panel sharing)
    const char* ab = getenv("CST_GB_ABL");                                             // timing ablations (tools/gemm_bench.py abl): results are wrong
    const int gn = gn0 | ((ab ? atoi(ab) : 0) << 16);
    static CstPerDevice attr_done;
    if (cst_first_on_device(attr_done)) {
        (void)hipFuncSetAttribute((const void*)cst_gemm_bf16_big_kernel<WR, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    dim3 grid((g.M / (128 * WR)) * (g.N / GB_T), 1, 1), block(256 * WR);
    if (cst_prof_on()) {
        hipEvent_t ea, eb;
        (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
        cst_prof_push_shape(ea, eb, 2.0 * g.M * g.N * g.K, 2.0 * ((double)g.M * g.K + (double)g.N * g.K) + (g.C ? 4.0 : 0.0) * g.M * g.N + (g.Cb ? 2.0 : 0.0) * g.M * g.N, 1,
                            g.M, g.N, g.K);
        hipExtLaunchKernelGGL((cst_gemm_bf16_big_kernel<WR, NST>), grid, block, lds, st, ea, eb, 0, g, gn);
    } else {
        hipLaunchKernelGGL((cst_gemm_bf16_big_kernel<WR, NST>), grid, block, lds, st, g, gn);
    }
    return 0;
}


#endif   // CST_BENCH_VARIANTS

template <int BM, int BN, int NSTAGE, bool TT = false, bool BF8 = false, int NW = 4>
static int bgemm_launch(const BGemmArgs& g, hipStream_t st) {
    const size_t lds = (size_t)NSTAGE * (BM * BROW + BN * (BF8 ? 64 : BROW));
    static CstPerDevice attr_done;
    if (lds > 64 * 1024 && cst_first_on_device(attr_done)) {
        (void)hipFuncSetAttribute((const void*)cst_gemm_bf16_kernel<BM, BN, NSTAGE, TT, BF8, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    dim3 grid(cst_div_up(g.M, BM) * cst_div_up(g.N, BN), g.splits, g.A2 ? 2 : 1), block(64 * NW);
    if (cst_prof_on()) {
        hipEvent_t ea, eb;
        (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
        // TT products are recorded with a negative K (C = A^T B: the contraction index is the row index of both operands)
        cst_prof_push_shape(ea, eb, 2.0 * g.M * g.N * g.K, 2.0 * ((double)g.M * g.K + (double)g.N * g.K) + (g.C ? 4.0 : 0.0) * g.M * g.N + (g.Cb ? 2.0 : 0.0) * g.M * g.N, 1,
                            g.M, g.N, TT ? -g.K : g.K);
        hipExtLaunchKernelGGL((cst_gemm_bf16_kernel<BM, BN, NSTAGE, TT, BF8, NW>), grid, block, lds, st, ea, eb, 0, g);
    } else {
        hipLaunchKernelGGL((cst_gemm_bf16_kernel<BM, BN, NSTAGE, TT, BF8, NW>), grid, block, lds, st, g);
    }
    return 0;
}

// Tile / split-K choice of cst_gemm_bf16 (and of cst_gemm_bf16_w8, which pins 64 x 128), shared with cst_gemm_bf16_workspace_floats.
// tools/gemm_bench.py bf16nt / enc: the 2-stage ring with two or three workgroups per CU beats the deeper ones; 64x128 tiles when they
// alone give >= 256 workgroups, 128x128 + split-K for long-K products with few output tiles, 64x128 (+ split) otherwise.
// Round quantisation: 64x128 tiles run 3 workgroups per CU (768 slots), 128x128 tiles 2 (512 slots).  At equal round efficiency the
// smaller tile wins; when the 128x128 grid fills its slots markedly better -- 9216 x 768 (432 of 512 against 864 of 1536),
// 4608 x 1536 -- it is 15-30 % faster.  `tile` arrives with the ring bits cleared (64 / 128 force a tile).
static void bgemm_plan(int M, int N, int K, int tile, int splitk, bool has_ws, long ws_floats, int* use_big_out, int* splits_out, int* kps_out) {
    const long big = (long)cst_div_up(M, 128) * cst_div_up(N, 128);
    const long mid = (long)cst_div_up(M, 64) * cst_div_up(N, 128);
    int use_big, splits = 1;
    if (tile == 128) use_big = 1;
    else if (tile == 64) use_big = 0;
    else {
        // long-K products with few output tiles (the vocabulary-side dgrads, K = 10 048: 4608 x 512 and 4608 x 1024): one or two 64 x 128
        // workgroups per CU walking 157 K-tiles one tile ahead are latency-bound; 128 x 128 tiles (+ the split below when they are fewer
        // than 192) measured 92 -> 58 us and 146 -> 115 us (tools/splitk_probe.py)
        use_big = (mid < 256 && K >= 2048) || (K >= 4096 && big <= 512);
        if (!use_big && splitk <= 1 && mid >= 256 && big >= 256) {
            // round 3 (tools/gemm_bench.py enc, tools/blas_reference.py): once both tilings fill the chip the 128 x 128 / 8-wave form is the
            // faster one (9216 x 2304 x 768: 47.8 against 52.4 us, 4608 x 10000 x 768: 96 against 107, 9216 x 1536 x 512: 25.7 against 27.4)
            // EXCEPT when its last round is nearly empty -- 576 tiles on 512 slots (4608 x 2048 x 768 / x 512): 27.9 against 25.9
            const double e128 = (double)big / (double)(cst_div_up(big, 512) * 512);
            if (e128 >= 0.6) use_big = 1;
        }
    }
    const long tiles = use_big ? big : mid;
    if (splitk > 1) splits = splitk;
    else if (splitk == 0 && has_ws && tiles < 192 && K >= 512) {
        splits = (int)((384 + tiles - 1) / tiles);
        if (splits > K / 256) splits = K / 256;
    }
    int kps = K;
    if (splits > 1) {
        kps = cst_div_up(cst_div_up(K, splits), 64) * 64;
        splits = cst_div_up(K, kps);
        while (splits > 1 && (long)splits * M * N > ws_floats) { kps += 64; splits = cst_div_up(K, kps); }
    }
    if (splits <= 1) { splits = 1; kps = K; }
    *use_big_out = use_big; *splits_out = splits; *kps_out = kps;
}

extern "C" long cst_gemm_bf16_workspace_floats(int M, int N, int K, int tile, int splitk) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int use_big, splits, kps;
    tile &= ~3;
    if (tile == 136) tile = 128;
    bgemm_plan(M, N, K, tile, splitk, true, 0x7fffffffffffffffL, &use_big, &splits, &kps);
    return splits > 1 ? (long)splits * M * N : 0;
}

// split-K choice of cst_gemm_bf16_tt (128 x 128 tiles only), shared with its workspace query
static void bgemm_tt_plan(int M, int N, int K, int splitk, bool has_ws, long ws_floats, int* splits_out, int* kps_out) {
    const long tiles = (long)cst_div_up(M, 128) * cst_div_up(N, 128);
    int splits = 1;
    if (splitk > 1) splits = splitk;
    else if (splitk == 0 && has_ws && tiles < 192 && K >= 512) {
        splits = (int)((384 + tiles - 1) / tiles);
        if (splits > K / 256) splits = K / 256;
    }
    int kps = K;
    if (splits > 1) {
        kps = cst_div_up(cst_div_up(K, splits), 64) * 64;
        splits = cst_div_up(K, kps);
        while (splits > 1 && (long)splits * M * N > ws_floats) { kps += 64; splits = cst_div_up(K, kps); }
    }
    if (splits <= 1) { splits = 1; kps = K; }
    *splits_out = splits; *kps_out = kps;
}

extern "C" long cst_gemm_bf16_tt_workspace_floats(int M, int N, int K, int splitk) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int splits, kps;
    bgemm_tt_plan(M, N, K, splitk, true, 0x7fffffffffffffffL, &splits, &kps);
    return splits > 1 ? (long)splits * M * N : 0;
}

// split-K choice of the recurrent products (cst_gemm_bf16_lstm / _lstm_bwd / _lstm_attn: 64 x 128 tiles, slabs only), shared with
// their workspace query
static void bgemm_lstm_plan(int M, int N, int K, int problems, int splitk, int* splits_out, int* kps_out) {
    const long tiles = (long)cst_div_up(M, 64) * cst_div_up(N, 128) * problems;
    int splits = splitk > 0 ? splitk : (int)((384 + tiles - 1) / tiles);
    if (splits > K / 128) splits = K / 128;
    if (splits < 1) splits = 1;
    const int kps = cst_div_up(cst_div_up(K, splits), 64) * 64;
    *splits_out = cst_div_up(K, kps); *kps_out = kps;
}

/* N = 4H gate columns; problems = 2 when both encoder directions go in one launch */
extern "C" long cst_gemm_bf16_lstm_workspace_floats(int M, int N, int K, int problems, int splitk) {
    if (M <= 0 || N <= 0 || K <= 0 || problems <= 0) return 0;
    int splits, kps;
    bgemm_lstm_plan(M, N, K, problems, splitk, &splits, &kps);
    return (long)problems * splits * M * N;
}

static int bgemm_entry(const void* A, long lda, const void* B, long ldb,
                       float* C, long ldc, void* Cb, long ldcb, int M, int N, int K,
                       const float* bias, const float* addend, long ldadd, const void* aux, long ldaux,
                       int act, float gate_scale, float alpha, int accumulate,
                       float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                       int tile, int splitk, float* workspace, long workspace_floats, void* stream, unsigned long long* amax) {
    CST_REQUIRE(A && B && (C || Cb), "cst_gemm_bf16: null operand");
    CST_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, "cst_gemm_bf16: K=%d must be a positive multiple of 64 (zero-padded operands)", K);
    CST_REQUIRE(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0, "cst_gemm_bf16: lda/ldb must be >= K and multiples of 8");
    CST_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0, "cst_gemm_bf16: operands must be 16-byte aligned");
    CST_REQUIRE(!C || ldc >= N, "cst_gemm_bf16: ldc < N");
    CST_REQUIRE(!Cb || ldcb >= N, "cst_gemm_bf16: ldcb < N");
    CST_REQUIRE(act >= 0 && act <= 4 && (act < 3 || aux), "cst_gemm_bf16: bad activation / missing aux");
    BGemmArgs g{};                                  // value-initialised: every pointer the entry point does not set is null
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.A2 = nullptr; g.B2 = nullptr; g.C = C; g.Cb = (bf16_t*)Cb;
    g.bias = bias; g.addend = addend; g.aux = (const bf16_t*)aux;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldcb = ldcb; g.ldadd = ldadd; g.ldaux = ldaux;
    g.M = M; g.N = N; g.K = K; g.act = act; g.alpha = alpha; g.gate_scale = gate_scale; g.accumulate = accumulate;
    g.slab_only = 0;
    g.amax = amax;
#ifdef CST_BENCH_VARIANTS
    { const char* ab = getenv("CST_GB_ABL"); g.abl = ab ? atoi(ab) : 0; }       // timing ablations (tools/gemm_bench.py abl): WRONG results
#else
    g.abl = 0;
#endif
    { static const int gn_env = getenv("CST_GEMM_GN") ? atoi(getenv("CST_GEMM_GN")) : 0; g.gn = gn_env; }
    g.drop = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)M * N);
    // tile / ring / split choice (tools/gemm_bench.py bf16nt): the 2-stage ring with two workgroups per
    // CU beats the 3-stage one at these sizes; 64x128 tiles when they alone give >= 256 workgroups,
    // 128x128 + split-K for long-K products with few output tiles (wgrad), 64x128 (+split) otherwise.
    // round 4 (gemm_pp.hip): the big-tile ping-pong kernel takes the large whole-K products (the encoder-layer forward and dgrad products)
    // unless a tile code asks for the tile kernels (64, 128, 136, 999 = "never").  Tile code 1000 + cfg forces one of its builds.  Which
    // build -- or the tile kernels below after all -- runs a shape is measured on the shape's first eager call: the tile kernels enter
    // that comparison through `again`, this same entry point with tile code 999.
    {
        int pp_force = 0;
        if (tile >= 1000) { pp_force = tile - 1000; tile = 0; }
        else if (tile != 0) pp_force = -1;
        if (tile == 999) tile = 0;
        if (pp_force >= 0 && splitk <= 1 && !amax) {
            struct Again {
                const void* A; long lda; const void* B; long ldb; float* C; long ldc; void* Cb; long ldcb; int M, N, K;
                const float* bias; const float* addend; long ldadd; const void* aux; long ldaux; int act; float gate_scale, alpha; int accumulate;
                float drop_p; uint32_t drop_seed, drop_stream; const void* drop_seed_dev; float* workspace; long workspace_floats; void* stream;
                static int run(void* p) {
                    const Again& a = *static_cast<const Again*>(p);
                    return bgemm_entry(a.A, a.lda, a.B, a.ldb, a.C, a.ldc, a.Cb, a.ldcb, a.M, a.N, a.K, a.bias, a.addend, a.ldadd, a.aux, a.ldaux, a.act, a.gate_scale,
                                       a.alpha, a.accumulate, a.drop_p, a.drop_seed, a.drop_stream, a.drop_seed_dev, 999, 0, a.workspace, a.workspace_floats, a.stream, nullptr);
                }
            } again{A, lda, B, ldb, C, ldc, Cb, ldcb, M, N, K, bias, addend, ldadd, aux, ldaux, act, gate_scale, alpha, accumulate,
                    drop_p, drop_seed, drop_stream, drop_seed_dev, workspace, workspace_floats, stream};
            g.splits = 1; g.k_per_split = K; g.slab = nullptr;
            const int rc = bgemm_pp_try(g, pp_force, (hipStream_t)stream, splitk == 0 ? &Again::run : nullptr, &again);
            if (rc > 0) { CST_LAUNCH_CHECK("cst_gemm_bf16 (ping-pong kernel)"); return CST_OK; }
        }
    }
    int ring = tile & 3;                         // tile code + 1 / + 2: force the 3- / 4-stage ring
    tile &= ~3;
    // 128 x 128 tiles run on 8 waves (sixteen DMA-issuing waves per CU) unless the 4-wave form is asked for: 0.15 ms per step at the
    // headline workload (25.48 against 25.63 ms, two A/B pairs in one session); CST_GEMM_W4 = 1 keeps the A/B switch
    static const bool w8_auto = getenv("CST_GEMM_W4") == nullptr;
    bool w8 = tile == 136;                       // 136: 128 x 128 tile on 8 waves
    if (w8) tile = 128;
    int use_big, splits;
    bgemm_plan(M, N, K, tile, splitk, workspace != nullptr, workspace_floats, &use_big, &splits, &g.k_per_split);
    const long tiles = use_big ? (long)cst_div_up(M, 128) * cst_div_up(N, 128) : (long)cst_div_up(M, 64) * cst_div_up(N, 128);
    CST_REQUIRE(splits == 1 || workspace, "cst_gemm_bf16: split-K needs a workspace");
    g.splits = splits; g.slab = workspace;
    hipStream_t st = (hipStream_t)stream;
#ifndef CST_BENCH_VARIANTS
    if (tile == 256 || tile == 252 || tile == 248) tile = 0;     // bench-only kernels: not in this build
    ring = 0;
#else
    // 256 x 256 / 8-wave kernel: tile code 256 forces it; by itself it takes the products whose 256^2 tiles fill at least
    // half the chip and at most one round of it, or several rounds (measured rule, tools/gemm_bench.py bf16nt)
    {
        static const int big_mode = getenv("CST_GEMM_BIG") ? atoi(getenv("CST_GEMM_BIG")) : 0;      // 0 only on request, 1 auto, 2 whenever legal
        // tile code 256: 256 x 256 / 8 waves; 252 (= 256 - 4, ring bits clear): 128 x 256 / 4 waves, two workgroups per CU
        const long t256 = (long)(M / GB_T) * (N / GB_T);
        const bool want = tile == 256 || tile == 252 || (tile == 0 && splitk <= 1 && big_mode == 2) ||
                          (tile == 0 && splitk <= 1 && big_mode == 1 && N >= 1024 && K >= 256 && t256 >= 128);
        if (tile == 248 && splits == 1 && bgemm_lc_ok(g)) {          // 248: loader / consumer persistent kernel (256 x 128 tiles)
            bgemm_lc_launch(g, st);
            CST_LAUNCH_CHECK("cst_gemm_bf16 (loader/consumer)");
            return CST_OK;
        }
        if (tile == 248) tile = 0;
        const int wr = tile == 256 ? 2 : 1;
        if (want && splits == 1 && bgemm_big_ok(g, wr)) {
            if (wr == 2) bgemm_big_launch_t<2, 4>(g, st);
            else bgemm_big_launch_t<1, 3>(g, st);
            CST_LAUNCH_CHECK("cst_gemm_bf16 (256-wide tiles)");
            return CST_OK;
        }
        if (tile == 256 || tile == 252) { tile = 0; }
    }
    if (ring == 0 && getenv("CST_RING4") && !use_big && tiles * splits <= 256 && g.k_per_split >= 256) ring = 2;
#endif
    (void)tiles;
    if (use_big && (w8 || w8_auto) && ring == 0) bgemm_launch<128, 128, 2, false, false, 8>(g, st);
#ifdef CST_BENCH_VARIANTS
    else if (use_big && ring == 1) bgemm_launch<128, 128, 3>(g, st);
    else if (!use_big && ring == 2) bgemm_launch<64, 128, 4>(g, st);
    else if (!use_big && ring == 1) bgemm_launch<64, 128, 3>(g, st);
#endif
    else if (use_big) bgemm_launch<128, 128, 2>(g, st);
    else bgemm_launch<64, 128, 2>(g, st);
    CST_LAUNCH_CHECK("cst_gemm_bf16");
    if (splits > 1) {
        long mn = ((long)M * N + 3) / 4;          // four columns per thread on the vector path
        int rb = (int)((mn + 255) / 256); if (rb > 2048) rb = 2048;
        static const bool no_batch = getenv("CST_REDUCE_SERIAL") != nullptr;      // A/B switch for tools/splitk_bench_bf16.py
        if (g.splits >= 5 && !no_batch) hipLaunchKernelGGL(cst_gemm_bf16_reduce<true>, dim3(rb), dim3(256), 0, st, g);
        else hipLaunchKernelGGL(cst_gemm_bf16_reduce<false>, dim3(rb), dim3(256), 0, st, g);
        CST_LAUNCH_CHECK("cst_gemm_bf16_reduce");
    }
    return CST_OK;
}

extern "C" int cst_gemm_bf16(const void* A, long lda, const void* B, long ldb,
                             float* C, long ldc, void* Cb, long ldcb, int M, int N, int K,
                             const float* bias, const float* addend, long ldadd, const void* aux, long ldaux,
                             int act, float gate_scale, float alpha, int accumulate,
                             float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                             int tile, int splitk, float* workspace, long workspace_floats, void* stream) {
    return bgemm_entry(A, lda, B, ldb, C, ldc, Cb, ldcb, M, N, K, bias, addend, ldadd, aux, ldaux, act, gate_scale, alpha, accumulate,
                       drop_p, drop_seed, drop_stream, drop_seed_dev, tile, splitk, workspace, workspace_floats, stream, nullptr);
}

extern "C" int cst_argmax_groups() { return CST_AMAX_GROUPS; }

// C[M, N] = A[M, K] . B[N, K]^T (fp32, no epilogue operands) and, per row, the packed arg-max word of bgemm_pack_max folded into
// amax_packed[group][m] with atomic max: the caller zeroes the words first and may spread one row's columns over several calls.  No K split
// (the epilogue that sees whole sums must be the GEMM's own).  Decoder: the fn_2 product of one decode step (rnn.py:80) whose arg-max
// feeds the next step (rnn.py:83-92); decode.hip.
extern "C" int cst_gemm_bf16_argmax(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                                    void* amax_packed, void* stream) {
    CST_REQUIRE(C && amax_packed && (((uintptr_t)amax_packed) & 7) == 0, "cst_gemm_bf16_argmax: null / misaligned output");
    return bgemm_entry(A, lda, B, ldb, C, ldc, nullptr, 0, M, N, K, nullptr, nullptr, 0, nullptr, 0, 0, 1.f, 1.f, 0,
                       0.f, 0, 0, nullptr, 64, 1, nullptr, 0, stream, (unsigned long long*)amax_packed);
}

// ---- grouped weight gradients: cst_gemm_bf16_tt calls between _group_begin and _group_end are collected and
// launched together by _group_end.  A group whose tiles would not fill the chip is launched product by product, as without a group.
struct TtDeferred { BTtProblem p; int splitk; float* ws; long ws_floats; };
// ONE group per process: it is opened inside a backward pass (autograd's device thread) and may be closed by the thread that started
// that pass; this host code issues GPU work from one thread at a time.
static struct { bool open; int n; TtDeferred d[CST_TT_GROUP_MAX]; int* counters; int ncounters; int last_splits; } tt_group = {false, 0, {}, nullptr, 0, 0};

static int tt_single(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                     int accumulate, int splitk, float* workspace, long workspace_floats, void* stream);

static int tt_group_env(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
static int tt_group_min_tiles() { static const int v = tt_group_env("CST_TT_GROUP_MIN", 224); return v; }     // 0: never group
static int tt_group_forced_splits() { static const int v = tt_group_env("CST_TT_GROUP_SPLITS", 0); return v; }   // 0: the rule below

// K splits of a grouped launch (tools/tt_group_probe.py, CST_TT_GROUP_SPLITS = 1 / 2 / 3).  Two workgroups fit a CU and a CU with two
// runs 1.6x as fast as a CU with one, so a group of up to 256 tiles (the d = 512 layers: 192) runs S = 2 workgroups per tile, all
// resident at once: 107 -> 83 us (9216 tokens), 405 -> 299 us (30720).  Above that the partial-tile round trip through memory costs
// more than the better balance returns (336 tiles, 9216 tokens: 135 us whole-K, 163 us with S = 2, 169 us with S = 3).
static int tt_group_splits(long tiles, int kmin, long sum_mn, long ws_floats, int ncounters) {
    const int forced = tt_group_forced_splits();
    const int S = forced > 0 ? forced : (2 * tiles <= 512 ? 2 : 1);
    if (S <= 1 || S > 4 || S * sum_mn > ws_floats || tiles > ncounters || kmin / S < 1024) return 1;
    return S;
}

static int tt_group_flush(hipStream_t st) {
    const int n = tt_group.n;
    tt_group.n = 0;
    if (n == 0) return CST_OK;
    BTtGroup q{};
    long tiles = 0, sum_mn = 0;
    double flops = 0, bytes = 0;
    int kmax = 0, kmin = 0x7fffffff;
    bool one_ws = true;
    for (int i = 0; i < n; ++i) {
        const BTtProblem& t = tt_group.d[i].p;
        q.p[i] = t;
        tiles += (long)cst_div_up(t.M, 128) * cst_div_up(t.N, 128);
        sum_mn += (long)t.M * t.N;
        flops += 2.0 * t.M * t.N * t.K;
        bytes += 2.0 * ((double)t.M * t.K + (double)t.N * t.K) + 4.0 * t.M * t.N;
        kmax = t.K > kmax ? t.K : kmax;
        kmin = t.K < kmin ? t.K : kmin;
        one_ws = one_ws && tt_group.d[i].ws == tt_group.d[0].ws && tt_group.d[i].ws_floats == tt_group.d[0].ws_floats;
    }
    // whole-K workgroups need enough tiles to occupy the CUs: 336 tiles (d = 768 layers) always win; 192 tiles (d = 512) with whole-K
    // workgroups win while the contraction is short (9216 tokens: 127 -> 102 us, 4608: 90 -> 54, 15360: 186 -> 176) and lose to
    // per-product split-K when it is long (30720 tokens: 344 -> 399 us); with two workgroups per tile they win throughout (see
    // tt_group_splits) -- tools/tt_group_probe.py
    const int min_tiles = tt_group_min_tiles();
    float* ws = tt_group.d[0].ws;
    const int S = (one_ws && ws && tt_group.counters && (((uintptr_t)ws) & 15) == 0)
                      ? tt_group_splits(tiles, kmin, sum_mn, tt_group.d[0].ws_floats, tt_group.ncounters) : 1;
    const bool fills = tiles >= min_tiles || (min_tiles == 224 && ((S == 2 && tiles >= 128) || (tiles >= 160 && kmax <= 16384)));
    if (n == 1 || min_tiles <= 0 || !fills) {
        tt_group.last_splits = 0;
        for (int i = 0; i < n; ++i) {
            const TtDeferred& d = tt_group.d[i];
            const int rc = tt_single(d.p.A, d.p.lda, d.p.B, d.p.ldb, d.p.C, d.p.ldc, d.p.M, d.p.N, d.p.K, d.p.accumulate, d.splitk, d.ws, d.ws_floats, st);
            if (rc != CST_OK) return rc;
        }
        return CST_OK;
    }
    tt_group.last_splits = S;
    long items = 0, off = 0;
    for (int i = 0; i < n; ++i) {
        BTtProblem& t = q.p[i];
        q.start[i] = (int)items;
        items += (long)cst_div_up(t.M, 128) * cst_div_up(t.N, 128) * S;
        t.kps = S > 1 ? cst_div_up(cst_div_up(t.K, S), 64) * 64 : t.K;        // (S - 1) kps < K: K >= 1024 S
        t.slab = S > 1 ? ws + off : nullptr;
        t.cvec = (((uintptr_t)t.C) & 15) == 0 && t.ldc % 4 == 0;
        off += S * (long)t.M * t.N;
    }
    q.start[n] = (int)items; q.n = n; q.splits = S; q.counters = tt_group.counters;
    constexpr size_t lds = 2 * (128 * BROW + 128 * BROW);
    static CstPerDevice attr_done;
    if (lds > 64 * 1024 && cst_first_on_device(attr_done))
        (void)hipFuncSetAttribute((const void*)cst_gemm_bf16_tt_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (cst_prof_on()) {
        hipEvent_t ea, eb;
        (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
        // recorded as ONE product [sum of M_p N_p / 128] x 128 over the first problem's K (bench.py labels the 128-column TT shape a group)
        cst_prof_push_shape(ea, eb, flops, bytes, 1, (int)(flops / (256.0 * q.p[0].K)), 128, -q.p[0].K);
        hipExtLaunchKernelGGL(cst_gemm_bf16_tt_group_kernel, dim3((unsigned)items), dim3(256), lds, st, ea, eb, 0, q);
    } else {
        hipLaunchKernelGGL(cst_gemm_bf16_tt_group_kernel, dim3((unsigned)items), dim3(256), lds, st, q);
    }
    CST_LAUNCH_CHECK("cst_gemm_bf16_tt_group");
    return CST_OK;
}

extern "C" int cst_gemm_bf16_tt_group_begin(int* counters, int ncounters, void* /*stream: the products are launched by _group_end, on ITS stream*/) {
    CST_REQUIRE(!tt_group.open, "cst_gemm_bf16_tt_group_begin: a group is already open");
    CST_REQUIRE(ncounters >= 0 && (counters || ncounters == 0), "cst_gemm_bf16_tt_group_begin: %d counters at a null address", ncounters);
    tt_group.open = true; tt_group.n = 0; tt_group.counters = counters; tt_group.ncounters = counters ? ncounters : 0;
    return CST_OK;
}

/* how the last group was launched: 0 = product by product, S >= 1 = one kernel with S workgroups per tile */
extern "C" int cst_gemm_bf16_tt_group_last_splits(void) { return tt_group.last_splits; }

extern "C" int cst_gemm_bf16_tt_group_end(void* stream) {
    CST_REQUIRE(tt_group.open, "cst_gemm_bf16_tt_group_end: no group is open");
    tt_group.open = false;
    return tt_group_flush((hipStream_t)stream);
}

extern "C" int cst_gemm_bf16_tt(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                                int accumulate, int splitk, float* workspace, long workspace_floats, void* stream) {
    CST_REQUIRE(A && B && C, "cst_gemm_bf16_tt: null operand");
    CST_REQUIRE(M >= 8 && N >= 8 && M % 8 == 0 && N % 8 == 0 && K > 0 && K % 64 == 0,
                "cst_gemm_bf16_tt: M=%d, N=%d must be multiples of 8 and K=%d a multiple of 64", M, N, K);
    CST_REQUIRE(lda >= M && ldb >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc >= N && (((uintptr_t)A | (uintptr_t)B) & 15) == 0,
                "cst_gemm_bf16_tt: operands must be 16-byte aligned, leading dimensions multiples of 8");
    if (tt_group.open && splitk >= 0 && splitk <= 1) {     // (an explicit split-K request, or splitk < 0 = "now, split as you see fit", runs at once)
        if (tt_group.n == CST_TT_GROUP_MAX) {
            const int rc = tt_group_flush((hipStream_t)stream);
            if (rc != CST_OK) return rc;
        }
        TtDeferred& d = tt_group.d[tt_group.n++];
        d.p = BTtProblem{(const bf16_t*)A, (const bf16_t*)B, C, lda, ldb, ldc, M, N, K, accumulate, nullptr, K, 0};
        d.splitk = splitk; d.ws = workspace; d.ws_floats = workspace_floats;
        return CST_OK;
    }
    return tt_single(A, lda, B, ldb, C, ldc, M, N, K, accumulate, splitk < 0 ? 0 : splitk, workspace, workspace_floats, stream);
}

static int tt_single(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                     int accumulate, int splitk, float* workspace, long workspace_floats, void* stream) {
    BGemmArgs g{};                                  // value-initialised: every pointer the entry point does not set is null
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.A2 = nullptr; g.B2 = nullptr; g.C = C; g.Cb = nullptr;
    g.bias = nullptr; g.addend = nullptr; g.aux = nullptr;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldcb = 0; g.ldadd = 0; g.ldaux = 0;
    g.M = M; g.N = N; g.K = K; g.act = 0; g.alpha = 1.f; g.gate_scale = 1.f; g.accumulate = accumulate;
    g.slab_only = 0;
    g.drop = cst_make_drop(0.f, 0, 0, nullptr, 0);
    int splits, kps;
    bgemm_tt_plan(M, N, K, splitk, workspace != nullptr, workspace_floats, &splits, &kps);
    CST_REQUIRE(splits == 1 || workspace, "cst_gemm_bf16_tt: split-K needs a workspace");
    g.splits = splits; g.k_per_split = kps; g.slab = workspace;
    hipStream_t st = (hipStream_t)stream;
    bgemm_launch<128, 128, 2, true>(g, st);
    CST_LAUNCH_CHECK("cst_gemm_bf16_tt");
    if (splits > 1) {
        long mn = ((long)M * N + 3) / 4;
        int rb = (int)((mn + 255) / 256); if (rb > 2048) rb = 2048;
        static const bool no_batch = getenv("CST_REDUCE_SERIAL") != nullptr;      // A/B switch for tools/splitk_bench_bf16.py
        if (g.splits >= 5 && !no_batch) hipLaunchKernelGGL(cst_gemm_bf16_reduce<true>, dim3(rb), dim3(256), 0, st, g);
        else hipLaunchKernelGGL(cst_gemm_bf16_reduce<false>, dim3(rb), dim3(256), 0, st, g);
        CST_LAUNCH_CHECK("cst_gemm_bf16_reduce");
    }
    return CST_OK;
}

// =============================================================================================
// W8A16: fp8 (e4m3, OCP) weights with one fp32 scale per output channel, bf16 activations, fp32 accumulation
// (BASELINE configs[4]).  cst_cast_fp8_rows quantises W [R, C] row by row: scale[r] = max|W[r, :]| / 448, q = rne(W / scale)
// (448 = largest e4m3 magnitude), columns C .. ldo-1 are written as 0 so that ldo can be the 64-padded K of the GEMM.
// The forward products use the copy of W itself (rows = output channels), the dgrad products a second copy of W^T quantised
// along ITS rows (the layer's input features): each product's scale then sits on the GEMM's output column, where the epilogue
// applies it.  Weight-gradient products involve activations only and stay bf16.
// =============================================================================================
__global__ __launch_bounds__(256) void cast_fp8_rows_kernel(const float* __restrict__ W, long ldw, long colstride, int R, int C,
                                                            unsigned char* __restrict__ out, long ldo, float* __restrict__ scale) {
    __shared__ float red[16];
    const int r = blockIdx.x;
    const float* row = W + (long)r * ldw;                  // element (r, c) = row[c * colstride]: colstride != 1 quantises a transposed view
    float amax = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) amax = fmaxf(amax, fabsf(row[(long)c * colstride]));
    amax = block_max(amax, red);
    const float sc = amax > 0.f ? amax / 448.f : 1.f;
    const float inv = 1.f / sc;
    if (threadIdx.x == 0) scale[r] = sc;
    for (int c4 = threadIdx.x * 4; c4 < ldo; c4 += 1024) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = c4 + e;
            v[e] = c < C ? fminf(fmaxf(row[(long)c * colstride] * inv, -448.f), 448.f) : 0.f;
        }
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
        *reinterpret_cast<int*>(out + (long)r * ldo + c4) = pk;
    }
}

extern "C" int cst_cast_fp8_rows(const float* W, long ldw, long colstride, int R, int C, void* out, long ldo, float* scale, void* stream) {
    CST_REQUIRE(W && out && scale && R > 0 && C > 0 && ldo >= C && ldo % 16 == 0 && colstride >= 1, "cst_cast_fp8_rows: bad arguments (ldo must be a multiple of 16 >= C)");
    CST_REQUIRE(((uintptr_t)out & 15) == 0, "cst_cast_fp8_rows: output must be 16-byte aligned");
    hipLaunchKernelGGL(cast_fp8_rows_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, W, ldw, colstride, R, C, (unsigned char*)out, ldo, scale);
    CST_LAUNCH_CHECK("cst_cast_fp8_rows");
    return CST_OK;
}

// C / Cb [M, N] = epi(alpha * bscale[n] * A[M, K] . Bq[N, K]^T): A bf16 (K padded to 64), Bq fp8 e4m3 [N, ldb bytes], everything
// else as cst_gemm_bf16 (64 x 128 tiles, 2-stage ring, split-K through the workspace).
extern "C" int cst_gemm_bf16_w8(const void* A, long lda, const void* Bq, long ldb, const float* bscale,
                                float* C, long ldc, void* Cb, long ldcb, int M, int N, int K,
                                const float* bias, const float* addend, long ldadd, const void* aux, long ldaux,
                                int act, float gate_scale, float alpha, int accumulate,
                                float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                                int splitk, float* workspace, long workspace_floats, void* stream) {
    CST_REQUIRE(A && Bq && bscale && (C || Cb), "cst_gemm_bf16_w8: null operand");
    CST_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, "cst_gemm_bf16_w8: K=%d must be a positive multiple of 64 (zero-padded operands)", K);
    CST_REQUIRE(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 16 == 0, "cst_gemm_bf16_w8: lda (elements) / ldb (bytes) must be >= K, multiples of 8 / 16");
    CST_REQUIRE((((uintptr_t)A | (uintptr_t)Bq) & 15) == 0, "cst_gemm_bf16_w8: operands must be 16-byte aligned");
    CST_REQUIRE((!C || ldc >= N) && (!Cb || ldcb >= N), "cst_gemm_bf16_w8: ldc / ldcb < N");
    CST_REQUIRE(act >= 0 && act <= 4 && (act < 3 || aux), "cst_gemm_bf16_w8: bad activation / missing aux");
    BGemmArgs g{};
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)Bq; g.C = C; g.Cb = (bf16_t*)Cb;
    g.bias = bias; g.addend = addend; g.aux = (const bf16_t*)aux; g.bscale = bscale;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldcb = ldcb; g.ldadd = ldadd; g.ldaux = ldaux;
    g.M = M; g.N = N; g.K = K; g.act = act; g.alpha = alpha; g.gate_scale = gate_scale; g.accumulate = accumulate;
    g.drop = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)M * N);
    int use_big_unused, splits;
    bgemm_plan(M, N, K, 64, splitk, workspace != nullptr, workspace_floats, &use_big_unused, &splits, &g.k_per_split);       // 64 x 128 tiles only
    CST_REQUIRE(splits == 1 || workspace, "cst_gemm_bf16_w8: split-K needs a workspace");
    g.splits = splits; g.slab = workspace;
    hipStream_t st = (hipStream_t)stream;
    bgemm_launch<64, 128, 2, false, true>(g, st);
    CST_LAUNCH_CHECK("cst_gemm_bf16_w8");
    if (splits > 1) {
        long mn = ((long)M * N + 3) / 4;
        int rb = (int)((mn + 255) / 256); if (rb > 2048) rb = 2048;
        if (g.splits >= 5) hipLaunchKernelGGL(cst_gemm_bf16_reduce<true>, dim3(rb), dim3(256), 0, st, g);
        else hipLaunchKernelGGL(cst_gemm_bf16_reduce<false>, dim3(rb), dim3(256), 0, st, g);
        CST_LAUNCH_CHECK("cst_gemm_bf16_reduce");
    }
    return CST_OK;
}

// =============================================================================================
// Recurrent step of nn.LSTM (rnn.py:25-33; gate order i, f, g, o) in two launches, forward and backward,
// for one problem or for two independent problems of one shape (the two directions of the encoder):
//   forward   gates = h W_hh^T (+ bias) (+ input projection); cell            (cst_gemm_bf16_lstm)
//   backward  dh = dgates_next W_hh; cell backward of the step before         (cst_gemm_bf16_lstm_bwd)
// The GEMM kernel leaves its split-K partial products in the workspace; the second kernel sums them in
// split order and applies the cell, one thread per 4 hidden units.  The separate reduce and cell
// launches and the round trip of the pre-activations / dh disappear from every recurrent step.
// =============================================================================================
struct LstmEpi {
    const float* bias; const float* addend;
    float* gates; const float* c_prev; float* h_out; float* c_out; float* h_out2;
    bf16_t* hb; bf16_t* hb2;
};
struct LstmEpi2 {
    LstmEpi e[2];
    const float* slab; int splits; int M, H;
    long ldadd, ldg, ldcp, ldh, ldc, ldh2, ldhb, ldhb2;
};

__device__ __forceinline__ float lstm_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ void f4_add(float (&a)[4], const float* p) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    a[0] += t.x; a[1] += t.y; a[2] += t.z; a[3] += t.w;
}
__device__ __forceinline__ void f4_store(float* p, const float (&a)[4]) { *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2], a[3]); }
__device__ __forceinline__ void bf4_store(bf16_t* p, const float (&a)[4]) {
    uint2 u;
    u.x = (uint32_t)f2bf16(a[0]) | ((uint32_t)f2bf16(a[1]) << 16);
    u.y = (uint32_t)f2bf16(a[2]) | ((uint32_t)f2bf16(a[3]) << 16);
    *reinterpret_cast<uint2*>(p) = u;
}

__global__ __launch_bounds__(256) void cst_gemm_bf16_lstm_reduce(LstmEpi2 q) {
    const LstmEpi& p = q.e[blockIdx.y];
    const int H4 = q.H >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= q.M * H4) return;
    const int m = idx / H4, u = (idx - m * H4) * 4;
    const long MN = (long)q.M * 4 * q.H;
    const float* slab = q.slab + (long)blockIdx.y * q.splits * MN;
    float pre[4][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const long off = (long)m * 4 * q.H + (long)g * q.H + u;
#pragma unroll
        for (int e = 0; e < 4; ++e) pre[g][e] = 0.f;
        for (int s = 0; s < q.splits; ++s) f4_add(pre[g], slab + s * MN + off);
        if (p.bias) f4_add(pre[g], p.bias + g * q.H + u);
        if (p.addend) f4_add(pre[g], p.addend + (long)m * q.ldadd + g * q.H + u);
    }
    float cp[4] = {0.f, 0.f, 0.f, 0.f};
    f4_add(cp, p.c_prev + (long)m * q.ldcp + u);
    float gi[4], gf[4], gg[4], go[4], c[4], h[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        gi[e] = lstm_sigmoid(pre[0][e]);
        gf[e] = lstm_sigmoid(pre[1][e]);
        gg[e] = tanhf(pre[2][e]);
        go[e] = lstm_sigmoid(pre[3][e]);
        c[e] = gf[e] * cp[e] + gi[e] * gg[e];
        h[e] = go[e] * tanhf(c[e]);
    }
    float* g = p.gates + (long)m * q.ldg + u;
    f4_store(g, gi); f4_store(g + q.H, gf); f4_store(g + 2 * q.H, gg); f4_store(g + 3 * q.H, go);
    f4_store(p.c_out + (long)m * q.ldc + u, c);
    f4_store(p.h_out + (long)m * q.ldh + u, h);
    if (p.h_out2) f4_store(p.h_out2 + (long)m * q.ldh2 + u, h);
    if (p.hb) bf4_store(p.hb + (long)m * q.ldhb + u, h);
    if (p.hb2) bf4_store(p.hb2 + (long)m * q.ldhb2 + u, h);
}

static bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// shared front end: shape checks, split choice and the slab-only GEMM launch for 1 or 2 problems
static int lstm_gemm_front(const char* who, BGemmArgs& g, const void* A, const void* B, const void* A2, const void* B2,
                           long lda, long ldb, int M, int N, int K, int splitk, float* workspace, long workspace_floats,
                           hipStream_t st) {
    CST_REQUIRE(A && B && workspace, "%s: null pointer", who);
    CST_REQUIRE((A2 == nullptr) == (B2 == nullptr), "%s: the second problem needs both operands", who);
    CST_REQUIRE(M > 0 && N > 0 && K > 0 && K % 64 == 0, "%s: K=%d must be a positive multiple of 64", who, K);
    CST_REQUIRE(lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && al16(A) && al16(B) && al16(A2) && al16(B2),
                "%s: operands must be 16-byte aligned with leading dimensions >= K, multiples of 8", who);
    g.A = (const bf16_t*)A; g.B = (const bf16_t*)B; g.A2 = (const bf16_t*)A2; g.B2 = (const bf16_t*)B2;
    g.C = nullptr; g.Cb = nullptr; g.bias = nullptr; g.addend = nullptr; g.aux = nullptr;
    g.lda = lda; g.ldb = ldb; g.ldc = 0; g.ldcb = 0; g.ldadd = 0; g.ldaux = 0;
    g.M = M; g.N = N; g.K = K; g.act = 0; g.alpha = 1.f; g.gate_scale = 1.f; g.accumulate = 0;
    g.drop = cst_make_drop(0.f, 0, 0, nullptr, 0);
    g.slab_only = 1;
    const int np = A2 ? 2 : 1;
    int splits, kps;
    bgemm_lstm_plan(M, N, K, np, splitk, &splits, &kps);
    CST_REQUIRE((long)np * splits * M * N <= workspace_floats, "%s: workspace too small", who);
    g.splits = splits; g.k_per_split = kps; g.slab = workspace;
    bgemm_launch<64, 128, 2>(g, st);
    CST_LAUNCH_CHECK(who);
    return CST_OK;
}

extern "C" int cst_gemm_bf16_lstm(const void* A, long lda, const void* B, long ldb, int M, int H, int K,
                                  const float* bias, const float* addend, long ldadd,
                                  float* gates, long ldg, const float* c_prev, long ldcp,
                                  float* h_out, long ldh, float* c_out, long ldc, float* h_out2, long ldh2,
                                  void* h_bf16, long ldhb, void* h_bf16_2, long ldhb2,
                                  const void* A2, const void* B2, const float* bias2, const float* addend2,
                                  float* gates2, const float* c_prev2, float* h_out_2, float* c_out2, float* h_out2_2,
                                  void* h_bf16_p2, void* h_bf16_2_p2,
                                  int splitk, float* workspace, long workspace_floats, void* stream) {
    CST_REQUIRE(gates && c_prev && h_out && c_out, "cst_gemm_bf16_lstm: null pointer");
    CST_REQUIRE(H > 0 && H % 4 == 0, "cst_gemm_bf16_lstm: H=%d must be a positive multiple of 4", H);
    CST_REQUIRE(!A2 || (gates2 && c_prev2 && h_out_2 && c_out2), "cst_gemm_bf16_lstm: second problem incomplete");
    const bool al = ((ldg | ldcp | ldh | ldc | ldadd | ldh2 | ldhb | ldhb2) % 4 == 0) &&
                    al16(gates) && al16(c_prev) && al16(h_out) && al16(c_out) && al16(bias) && al16(addend) && al16(h_out2) &&
                    al16(gates2) && al16(c_prev2) && al16(h_out_2) && al16(c_out2) && al16(bias2) && al16(addend2) && al16(h_out2_2) &&
                    al16(workspace) && ((((uintptr_t)h_bf16 | (uintptr_t)h_bf16_2 | (uintptr_t)h_bf16_p2 | (uintptr_t)h_bf16_2_p2) & 7) == 0);
    CST_REQUIRE(al, "cst_gemm_bf16_lstm: every row pointer must be 16-byte aligned (leading dimensions multiples of 4)");
    hipStream_t st = (hipStream_t)stream;
    BGemmArgs g{};                                  // value-initialised: every pointer the entry point does not set is null
    if (int rc = lstm_gemm_front("cst_gemm_bf16_lstm", g, A, B, A2, B2, lda, ldb, M, 4 * H, K, splitk, workspace, workspace_floats, st)) return rc;
    LstmEpi2 q;
    q.e[0] = LstmEpi{bias, addend, gates, c_prev, h_out, c_out, h_out2, (bf16_t*)h_bf16, (bf16_t*)h_bf16_2};
    q.e[1] = LstmEpi{bias2, addend2, gates2, c_prev2, h_out_2, c_out2, h_out2_2, (bf16_t*)h_bf16_p2, (bf16_t*)h_bf16_2_p2};
    q.slab = workspace; q.splits = g.splits; q.M = M; q.H = H;
    q.ldadd = ldadd; q.ldg = ldg; q.ldcp = ldcp; q.ldh = ldh; q.ldc = ldc; q.ldh2 = ldh2; q.ldhb = ldhb; q.ldhb2 = ldhb2;
    const int nthr = M * (H / 4);
    hipLaunchKernelGGL(cst_gemm_bf16_lstm_reduce, dim3((nthr + 255) / 256, A2 ? 2 : 1), dim3(256), 0, st, q);
    CST_LAUNCH_CHECK("cst_gemm_bf16_lstm_reduce");
    return CST_OK;
}

// ---- backward --------------------------------------------------------------------------------
struct LstmBwdEpi {
    const float* gates; const float* c_prev; const float* c_new; const float* dh_extra; const float* dc_in;
    float* dgates; float* dc_prev; bf16_t* dgb;
};
struct LstmBwdEpi2 {
    LstmBwdEpi e[2];
    float* extra[2]; long ldx; int n_extra;           // leading n_extra columns of the product: summed and stored as they are
    const float* slab; int splits; int M, H;
    long ldg, ldcp, ldcn, lddh, lddc, lddg, lddcp, lddgb;
};

__global__ __launch_bounds__(256) void cst_gemm_bf16_lstm_bwd_reduce(LstmBwdEpi2 q) {
    const LstmBwdEpi& p = q.e[blockIdx.y];
    const int H4 = q.H >> 2, X4 = q.n_extra >> 2;
    const int N = q.n_extra + q.H;
    const long MN = (long)q.M * N;
    const float* slab = q.slab + (long)blockIdx.y * q.splits * MN;
    int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= q.M * H4) {                               // the pass-through columns (decoder: d x_t next to d h_{t-1})
        idx -= q.M * H4;
        if (idx >= q.M * X4) return;
        const int m = idx / X4, c = (idx - m * X4) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < q.splits; ++s) f4_add(v, slab + s * MN + (long)m * N + c);
        f4_store(q.extra[blockIdx.y] + (long)m * q.ldx + c, v);
        return;
    }
    const int m = idx / H4, u = (idx - m * H4) * 4;
    float dht[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < q.splits; ++s) f4_add(dht, slab + s * MN + (long)m * N + q.n_extra + u);
    if (p.dh_extra) f4_add(dht, p.dh_extra + (long)m * q.lddh + u);
    float gi[4] = {0.f, 0.f, 0.f, 0.f}, gf[4] = {0.f, 0.f, 0.f, 0.f}, gg[4] = {0.f, 0.f, 0.f, 0.f}, go[4] = {0.f, 0.f, 0.f, 0.f};
    const float* g = p.gates + (long)m * q.ldg + u;
    f4_add(gi, g); f4_add(gf, g + q.H); f4_add(gg, g + 2 * q.H); f4_add(go, g + 3 * q.H);
    float cp[4] = {0.f, 0.f, 0.f, 0.f}, cn[4] = {0.f, 0.f, 0.f, 0.f}, dc[4] = {0.f, 0.f, 0.f, 0.f};
    f4_add(cp, p.c_prev + (long)m * q.ldcp + u);
    f4_add(cn, p.c_new + (long)m * q.ldcn + u);
    if (p.dc_in) f4_add(dc, p.dc_in + (long)m * q.lddc + u);
    float d0[4], d1[4], d2[4], d3[4], dcp[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {                     // same arithmetic as lstm_cell_bwd_kernel
        const float tc = tanhf(cn[e]);
        const float dct = dc[e] + dht[e] * go[e] * (1.f - tc * tc);
        d0[e] = dct * gg[e] * gi[e] * (1.f - gi[e]);
        d1[e] = dct * cp[e] * gf[e] * (1.f - gf[e]);
        d2[e] = dct * gi[e] * (1.f - gg[e] * gg[e]);
        d3[e] = dht[e] * tc * go[e] * (1.f - go[e]);
        dcp[e] = dct * gf[e];
    }
    float* dg = p.dgates + (long)m * q.lddg + u;
    f4_store(dg, d0); f4_store(dg + q.H, d1); f4_store(dg + 2 * q.H, d2); f4_store(dg + 3 * q.H, d3);
    if (p.dgb) {
        bf16_t* b = p.dgb + (long)m * q.lddgb + u;
        bf4_store(b, d0); bf4_store(b + q.H, d1); bf4_store(b + 2 * q.H, d2); bf4_store(b + 3 * q.H, d3);
    }
    f4_store(p.dc_prev + (long)m * q.lddcp + u, dcp);
}

extern "C" int cst_gemm_bf16_lstm_bwd(const void* A, long lda, const void* B, long ldb, int M, int H, int K,
                                      const float* gates, long ldg, const float* c_prev, long ldcp, const float* c_new, long ldcn,
                                      const float* dh_extra, long lddh, const float* dc_in, long lddc,
                                      float* dgates, long lddg, float* dc_prev, long lddcp, void* dgates_bf16, long lddgb,
                                      const void* A2, const void* B2, const float* gates2, const float* c_prev2, const float* c_new2,
                                      const float* dh_extra2, const float* dc_in2, float* dgates2, float* dc_prev2, void* dgates_bf16_2,
                                      int n_extra, float* extra_out, float* extra_out2, long ldx,
                                      int splitk, float* workspace, long workspace_floats, void* stream) {
    CST_REQUIRE(gates && c_prev && c_new && dgates && dc_prev, "cst_gemm_bf16_lstm_bwd: null pointer");
    CST_REQUIRE(n_extra >= 0 && n_extra % 4 == 0 && (n_extra == 0 || (extra_out && ldx >= n_extra && ldx % 4 == 0 && al16(extra_out) && al16(extra_out2))),
                "cst_gemm_bf16_lstm_bwd: bad pass-through columns (n_extra=%d)", n_extra);
    CST_REQUIRE(H > 0 && H % 4 == 0, "cst_gemm_bf16_lstm_bwd: H=%d must be a positive multiple of 4", H);
    CST_REQUIRE(!A2 || (gates2 && c_prev2 && c_new2 && dgates2 && dc_prev2), "cst_gemm_bf16_lstm_bwd: second problem incomplete");
    const bool al = ((ldg | ldcp | ldcn | lddh | lddc | lddg | lddcp | lddgb) % 4 == 0) &&
                    al16(gates) && al16(c_prev) && al16(c_new) && al16(dh_extra) && al16(dc_in) && al16(dgates) && al16(dc_prev) &&
                    al16(gates2) && al16(c_prev2) && al16(c_new2) && al16(dh_extra2) && al16(dc_in2) && al16(dgates2) && al16(dc_prev2) &&
                    al16(workspace) && ((((uintptr_t)dgates_bf16 | (uintptr_t)dgates_bf16_2) & 7) == 0);
    CST_REQUIRE(al, "cst_gemm_bf16_lstm_bwd: every row pointer must be 16-byte aligned (leading dimensions multiples of 4)");
    hipStream_t st = (hipStream_t)stream;
    BGemmArgs g{};                                  // value-initialised: every pointer the entry point does not set is null
    if (int rc = lstm_gemm_front("cst_gemm_bf16_lstm_bwd", g, A, B, A2, B2, lda, ldb, M, n_extra + H, K, splitk, workspace, workspace_floats, st)) return rc;
    LstmBwdEpi2 q;
    q.e[0] = LstmBwdEpi{gates, c_prev, c_new, dh_extra, dc_in, dgates, dc_prev, (bf16_t*)dgates_bf16};
    q.e[1] = LstmBwdEpi{gates2, c_prev2, c_new2, dh_extra2, dc_in2, dgates2, dc_prev2, (bf16_t*)dgates_bf16_2};
    q.extra[0] = extra_out; q.extra[1] = extra_out2; q.ldx = ldx; q.n_extra = n_extra;
    q.slab = workspace; q.splits = g.splits; q.M = M; q.H = H;
    q.ldg = ldg; q.ldcp = ldcp; q.ldcn = ldcn; q.lddh = lddh; q.lddc = lddc; q.lddg = lddg; q.lddcp = lddcp; q.lddgb = lddgb;
    const int nthr = M * ((H + n_extra) / 4);
    hipLaunchKernelGGL(cst_gemm_bf16_lstm_bwd_reduce, dim3((nthr + 255) / 256, A2 ? 2 : 1), dim3(256), 0, st, q);
    CST_LAUNCH_CHECK("cst_gemm_bf16_lstm_bwd_reduce");
    return CST_OK;
}

// =============================================================================================
// Decoder step front half in two launches (rnn.py:75-79): gates = [x_t | h_{t-1}] [W_ih | W_hh]^T + b, LSTM cell,
// single-query attention of h_t over the encoder states, dropout of the FFN input [h_t | a_t].  One workgroup per
// batch row sums that row's split-K partials, applies the cell (one thread per hidden unit), keeps h_t in LDS as the
// attention query and finishes with the attention of cst_dot_attn_fwd: the h_t round trip and one dependent launch
// per decode step disappear.
// =============================================================================================
struct LstmAttnEpi {
    const float* slab; int splits; int M, H;
    const float* bias; float* gates; long ldg; const float* c_prev; long ldcp;
    float* h_out; long ldh; float* c_out; long ldc; float* h_out2; long ldh2; bf16_t* hb2; long ldhb2;
    const float* mem; float* att; long ldo; float* p; int L; float scale;
    float* dropped; long lddrop; bf16_t* dropped_b; long lddropb; CstDrop drop;
};

__global__ __launch_bounds__(1024) void cst_gemm_bf16_lstm_attn_kernel(LstmAttnEpi q) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int D = q.H, L = q.L;
    float* ms = dsm;                 // [L][D] encoder states of this batch row
    float* qs = ms + L * D;          // [D]    h_t
    float* sc = qs + D;              // [64]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // the cell's operands first (split-K slabs, bias, c_{t-1}), then the memory tile: both sets of loads are in flight
    // together -- issued behind the tile's LDS stores they cost the kernel a second memory round trip
    float pre[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (tid < D) {
        const int u = tid;
        const long MN = (long)q.M * 4 * D;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const long off = (long)b * 4 * D + (long)g * D + u;
            const float s0 = q.slab[off];
            const float s1 = q.slab[(q.splits > 1 ? MN : 0) + off];          // unconditional load, select below: stays in flight with s0
            float a = ((q.bias ? q.bias[g * D + u] : 0.f) + s0) + (q.splits > 1 ? s1 : 0.f);
            for (int s = 2; s < q.splits; ++s) a += q.slab[s * MN + off];
            pre[g] = a;
        }
        cprev = q.c_prev[(long)b * q.ldcp + u];
    }
    // memory tile: four 16-byte loads in flight per thread
    {
        const float4* src = reinterpret_cast<const float4*>(q.mem + (long)b * L * D);
        float4* dst = reinterpret_cast<float4*>(ms);
        const int n4 = L * D / 4, bd = blockDim.x;
        for (int base = 0; base < n4; base += 4 * bd) {
            float4 t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) t[u] = src[min(base + u * bd + tid, n4 - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int e = base + u * bd + tid;
                if (e < n4) dst[e] = t[u];
            }
        }
    }
    if (tid < D) {
        const int u = tid;
        const float gi = lstm_sigmoid(pre[0]), gf = lstm_sigmoid(pre[1]), gg = tanhf(pre[2]), go = lstm_sigmoid(pre[3]);
        const float c = gf * cprev + gi * gg;
        const float h = go * tanhf(c);
        float* g = q.gates + (long)b * q.ldg + u;
        g[0] = gi; g[D] = gf; g[2 * D] = gg; g[3 * D] = go;
        q.c_out[(long)b * q.ldc + u] = c;
        q.h_out[(long)b * q.ldh + u] = h;
        if (q.h_out2) q.h_out2[(long)b * q.ldh2 + u] = h;
        if (q.hb2) q.hb2[(long)b * q.ldhb2 + u] = f2bf16(h);
        qs[u] = h;
    }
    __syncthreads();
    for (int j = w; j < L; j += (int)(blockDim.x >> 6)) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += qs[c] * ms[j * D + c];
        s = wave_sum(s);
        if (lane == 0) sc[j] = s * q.scale;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int j = 0; j < L; ++j) m = fmaxf(m, sc[j]);
    float sum = 0.f;
    for (int j = 0; j < L; ++j) sum += expf(sc[j] - m);
    __syncthreads();
    if (tid < L) {
        const float pj = expf(sc[tid] - m) / sum;
        sc[tid] = pj;
        q.p[(long)b * L + tid] = pj;
    }
    __syncthreads();
    const uint32_t dseed = q.drop.p > 0.f ? cst_drop_seed(q.drop) : 0u;
    if (tid < D) {
        const int c = tid;
        float o = 0.f;
        for (int j = 0; j < L; ++j) o += sc[j] * ms[j * D + c];
        q.att[(long)b * q.ldo + c] = o;
        if (q.dropped || q.dropped_b) {                 // dropout index space is the (B, 2D) matrix [h | a]
            const float mq = q.drop.p > 0.f ? cst_drop_mask(q.drop, dseed, (uint32_t)((long)b * 2 * D + c)) : 1.f;
            const float mo = q.drop.p > 0.f ? cst_drop_mask(q.drop, dseed, (uint32_t)((long)b * 2 * D + D + c)) : 1.f;
            if (q.dropped) {
                float* dr = q.dropped + (long)b * q.lddrop;
                dr[c] = qs[c] * mq;
                dr[D + c] = o * mo;
            }
            if (q.dropped_b) {
                bf16_t* db = q.dropped_b + (long)b * q.lddropb;
                db[c] = f2bf16(qs[c] * mq);
                db[D + c] = f2bf16(o * mo);
            }
        }
    }
}

extern "C" int cst_gemm_bf16_lstm_attn(const void* A, long lda, const void* B, long ldb, int M, int H, int K,
                                       const float* bias, float* gates, long ldg, const float* c_prev, long ldcp,
                                       float* h_out, long ldh, float* c_out, long ldc, float* h_out2, long ldh2, void* h_bf16_2, long ldhb2,
                                       const float* mem, int L, float* att_out, long ldo, float* p,
                                       float* dropped, long lddrop, void* dropped_bf16, long lddropb,
                                       float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                                       int splitk, float* workspace, long workspace_floats, void* stream) {
    CST_REQUIRE(gates && c_prev && h_out && c_out && mem && att_out && p, "cst_gemm_bf16_lstm_attn: null pointer");
    CST_REQUIRE(H > 0 && H % 4 == 0 && H <= 1024 && L > 0 && L <= 64, "cst_gemm_bf16_lstm_attn: H=%d (multiple of 4, <= 1024) or L=%d (<= 64) unsupported", H, L);
    CST_REQUIRE(((uintptr_t)mem & 15) == 0, "cst_gemm_bf16_lstm_attn: mem must be 16-byte aligned");
    const size_t lds = sizeof(float) * ((size_t)L * H + H + 64);
    CST_REQUIRE(lds <= 160 * 1024, "cst_gemm_bf16_lstm_attn: memory tile of %zu bytes exceeds the 160 KiB LDS", lds);
    hipStream_t st = (hipStream_t)stream;
    BGemmArgs g{};                                  // value-initialised: every pointer the entry point does not set is null
    if (int rc = lstm_gemm_front("cst_gemm_bf16_lstm_attn", g, A, B, nullptr, nullptr, lda, ldb, M, 4 * H, K, splitk, workspace, workspace_floats, st)) return rc;
    LstmAttnEpi q;
    q.slab = workspace; q.splits = g.splits; q.M = M; q.H = H;
    q.bias = bias; q.gates = gates; q.ldg = ldg; q.c_prev = c_prev; q.ldcp = ldcp;
    q.h_out = h_out; q.ldh = ldh; q.c_out = c_out; q.ldc = ldc; q.h_out2 = h_out2; q.ldh2 = ldh2; q.hb2 = (bf16_t*)h_bf16_2; q.ldhb2 = ldhb2;
    q.mem = mem; q.att = att_out; q.ldo = ldo; q.p = p; q.L = L; q.scale = 1.0f / sqrtf((float)H);
    q.dropped = dropped; q.lddrop = lddrop; q.dropped_b = (bf16_t*)dropped_bf16; q.lddropb = lddropb;
    q.drop = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)M * 2 * H);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)cst_gemm_bf16_lstm_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(cst_gemm_bf16_lstm_attn_kernel, dim3(M), dim3(1024), lds, st, q);
    CST_LAUNCH_CHECK("cst_gemm_bf16_lstm_attn_kernel");
    return CST_OK;
}
