// MFMA GEMM for gfx950:  C[M,N] = epilogue( alpha * op(A)[M,K] . op(B)[K,N] )
//
// Every dense contraction of the hot path goes through this kernel: QKV / out-proj / FFN
// projections of the encoder layers (mlm.py:20-22, match.py:18-20), the vocabulary
// projections (mlm.py:24, rnn.py:38), the soft-input embedding products (rnn.py:61,85,
// mlm.py:31, match.py:28, classifier.py:27, discriminator.py:39), the LSTM gate projections
// (rnn.py:25-33), the im2col'ed convolutions and the highway layer -- forward, dgrad and wgrad.
//
// Operand layouts (fp32 in HBM, row-major with a leading dimension):
//   A "K-major"  : A[m*lda + k]      (an activation matrix used as is)
//   A "MN-major" : A[k*lda + m]      (the same matrix used transposed: wgrad)
//   B "K-major"  : B[n*ldb + k]      (a torch Linear weight [out,in] used as is: forward)
//   B "MN-major" : B[k*ldb + n]      (the weight used transposed: dgrad; activations in wgrad)
// so no operand is ever transposed in HBM.  Tiles are staged global -> registers -> LDS as
// [row][k] images (k contiguous, XOR-swizzled 16-byte slots) whatever the HBM layout; MN-major
// tiles are transposed in registers on the way.  Two arithmetic modes:
//   bf16 : v_mfma_f32_16x16x32_bf16, operands rounded to bf16 while staging, fp32 accumulate
//   f32  : v_mfma_f32_16x16x4_f32, exact fp32 products (the parity mode)
// 256 threads = 4 waves in a 2x2 grid; tile BMxBN = 128x128 or 64x64; a K-tile is 128 bytes per
// row in LDS (BK = 64 in bf16 mode, 32 in f32 mode), double buffered, one barrier per K-tile.
#include "cst_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8_t;

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias;      // [N] or null
    const float* addend;    // [M, ldadd] or null (added before the activation)
    const float* aux;       // [M, ldaux] gate source for act 3/4
    long lda, ldb, ldc, ldadd, ldaux;
    long sA, sB, sC, sBias, sAdd, sAux;   // batch strides (elements)
    int M, N, K;
    int act;                // 0 none, 1 relu, 2 leaky(0.1), 3 gate aux>0 ? v*gate_scale : 0, 4 gate aux>0 ? v : 0.1 v
    int accumulate;         // C += result
    int vecA, vecB;         // 16-byte vector loads legal for this operand
    float alpha, gate_scale;
    CstDrop drop;           // epilogue dropout over the logical [M,N] index space
    int splits;             // split-K: blockIdx.y owns k in [y*k_per_split, (y+1)*k_per_split)
    int k_per_split;        // multiple of BK
    float* slab;            // [batch][splits][M][N] raw partial sums when splits > 1
};

// epilogue shared by the in-kernel path and the split-K reduction
__device__ __forceinline__ void gemm_epilogue_store(const GemmArgs& g, float* C, const float* bias, const float* addend,
                                                    const float* aux, uint32_t dseed, long bz, int m, int n, float acc) {
    float v = g.alpha * acc + (bias ? bias[n] : 0.f);
    if (addend) v += addend[(long)m * g.ldadd + n];
    if (g.act == 1) v = v > 0.f ? v : 0.f;
    else if (g.act == 2) v = v > 0.f ? v : 0.1f * v;
    else if (g.act == 3) v = aux[(long)m * g.ldaux + n] > 0.f ? v * g.gate_scale : 0.f;
    else if (g.act == 4) v = aux[(long)m * g.ldaux + n] > 0.f ? v : 0.1f * v;
    if (g.drop.p > 0.f) v *= cst_drop_mask(g.drop, dseed, (uint32_t)((long)m * g.N + n + bz * (long)g.M * g.N));
    float* cp = C + (long)m * g.ldc + n;
    if (g.accumulate) v += *cp;
    *cp = v;
}

__device__ __forceinline__ unsigned short f2bf(float f) {
    // round-to-nearest-even; NaN stays NaN via the hardware convert
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

// ---- LDS tile image -----------------------------------------------------------------------
// Every operand tile is stored as [rows][128 bytes]: 64 bf16 (bf16 mode, BK = 64) or 32 fp32
// (f32 mode, BK = 32) of consecutive k per row, i.e. 8 slots of 16 bytes.  Slot s of row r lives
// at physical slot s ^ (r & 7): with 128-byte rows every row starts on bank 0, and the XOR makes
// both the 16-byte staging stores (8 lanes = 8 slots of one row, or 8 consecutive rows of one
// slot) and the ds_read_b128 fragment reads (16 rows x 4 k-slots per wave) bank-conflict free.
template <bool F32> struct TileCfg { static constexpr int BK = 64; };
template <> struct TileCfg<true> { static constexpr int BK = 32; };
constexpr int ROW_BYTES = 128;

__device__ __forceinline__ int lds_off(int row, int slot) { return row * ROW_BYTES + ((slot ^ (row & 7)) << 4); }

template <bool F32>
__device__ __forceinline__ uint4 pack_slot(const float (&v)[8]) {
    uint4 u;
    if constexpr (F32) {        // only v[0..3] are meaningful
        u.x = __float_as_uint(v[0]); u.y = __float_as_uint(v[1]); u.z = __float_as_uint(v[2]); u.w = __float_as_uint(v[3]);
    } else {
        u.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        u.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        u.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
        u.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
    }
    return u;
}

// ---- global -> register -> LDS staging ---------------------------------------------------------
// One "item" = one 16-byte LDS slot = EPS consecutive k of one row (8 bf16 or 4 fp32).
// K-major operand : item idx -> (row = idx / 8, slot = idx % 8); the k run is contiguous in HBM
//                   (one or two float4 loads; 8 lanes cover 128..256 contiguous bytes of a row).
// MN-major operand: item idx -> (row = idx % ROWS, slot = idx / ROWS); for a fixed k the rows are
//                   contiguous in HBM, so each of the EPS scalar loads is coalesced across lanes
//                   (64 lanes x 4 B) and the transpose happens in registers.
template <int ROWS, bool F32, bool KMAJ, int NT>
struct Stage;

// K-major, bf16 image: an item is one float4 (4 consecutive k) -> half a slot (8 bytes).
// 16 consecutive lanes read one whole 256-byte row of the HBM tile; their 16 ds_write_b64 cover
// the whole 128-byte LDS row (every bank once).
template <int ROWS, int NT>
struct Stage<ROWS, false, true, NT> {
    static constexpr int PER_T = ROWS * 16 / NT;
    float4 v[PER_T];
    __device__ __forceinline__ void load(const float* __restrict__ P, long ld, int row0, int k0, int rows, int kend, int vec) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = threadIdx.x + NT * i;
            const int r = idx >> 4, k = k0 + (idx & 15) * 4;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < rows) {
                const float* p = P + (long)(row0 + r) * ld + k;
                if (k + 0 < kend) t.x = p[0];
                if (k + 1 < kend) t.y = p[1];
                if (k + 2 < kend) t.z = p[2];
                if (k + 3 < kend) t.w = p[3];
            }
            v[i] = t;
        }
    }
    // full K-tile, 16-byte addressable operand: every load is unconditional (rows past the edge
    // re-read the last valid row; their products land in output rows that are never stored), so
    // the compiler issues all of them back to back behind ONE wait -- a guarded load makes hipcc
    // branch around it and drain vmcnt at every join.
    __device__ __forceinline__ void load_full(const float* __restrict__ P, long ld, int row0, int k0, int rows) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = threadIdx.x + NT * i;
            const int r = min(row0 + (idx >> 4), rows - 1), k = k0 + (idx & 15) * 4;
            v[i] = *reinterpret_cast<const float4*>(P + (long)r * ld + k);
        }
    }
    __device__ __forceinline__ void store(char* S) const {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = threadIdx.x + NT * i;
            const int r = idx >> 4, q = idx & 15;
            uint2 u;
            u.x = (uint32_t)f2bf(v[i].x) | ((uint32_t)f2bf(v[i].y) << 16);
            u.y = (uint32_t)f2bf(v[i].z) | ((uint32_t)f2bf(v[i].w) << 16);
            *reinterpret_cast<uint2*>(S + lds_off(r, q >> 1) + (q & 1) * 8) = u;
        }
    }
};

// K-major, fp32 image: an item is one float4 = one slot; 8 lanes cover a 128-byte row.
template <int ROWS, int NT>
struct Stage<ROWS, true, true, NT> {
    static constexpr int PER_T = ROWS * 8 / NT;
    float4 v[PER_T];
    __device__ __forceinline__ void load(const float* __restrict__ P, long ld, int row0, int k0, int rows, int kend, int vec) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = threadIdx.x + NT * i;
            const int r = idx >> 3, k = k0 + (idx & 7) * 4;
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < rows) {
                const float* p = P + (long)(row0 + r) * ld + k;
                if (k + 0 < kend) t.x = p[0];
                if (k + 1 < kend) t.y = p[1];
                if (k + 2 < kend) t.z = p[2];
                if (k + 3 < kend) t.w = p[3];
            }
            v[i] = t;
        }
    }
    __device__ __forceinline__ void load_full(const float* __restrict__ P, long ld, int row0, int k0, int rows) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = threadIdx.x + NT * i;
            const int r = min(row0 + (idx >> 3), rows - 1), k = k0 + (idx & 7) * 4;
            v[i] = *reinterpret_cast<const float4*>(P + (long)r * ld + k);
        }
    }
    __device__ __forceinline__ void store(char* S) const {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = threadIdx.x + NT * i;
            *reinterpret_cast<float4*>(S + lds_off(idx >> 3, idx & 7)) = v[i];
        }
    }
};

// MN-major (either image): an item is one 16-byte slot = EPS consecutive k of one row; for a
// fixed k the rows are contiguous in HBM, so each of the EPS scalar loads is coalesced across the
// 64 lanes (256 bytes) and the transpose happens in registers; 8 consecutive lanes (rows) then
// write 8 distinct physical slots.
template <int ROWS, bool F32, int NT>
struct Stage<ROWS, F32, false, NT> {
    static constexpr int EPS = F32 ? 4 : 8;
    static constexpr int PER_T = ROWS * 8 / NT;
    float v[PER_T][8];
    __device__ __forceinline__ void load(const float* __restrict__ P, long ld, int row0, int k0, int rows, int kend, int vec) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = threadIdx.x + NT * i;
            const int r = idx % ROWS, k = k0 + (idx / ROWS) * EPS;
            const bool rok = row0 + r < rows;
            const float* p = P + (long)k * ld + row0 + r;
#pragma unroll
            for (int e = 0; e < EPS; ++e) v[i][e] = (rok && k + e < kend) ? p[(long)e * ld] : 0.f;
        }
    }
    __device__ __forceinline__ void load_full(const float* __restrict__ P, long ld, int row0, int k0, int rows) {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = threadIdx.x + NT * i;
            const int r = min(row0 + idx % ROWS, rows - 1), k = k0 + (idx / ROWS) * EPS;
            const float* p = P + (long)k * ld + r;
#pragma unroll
            for (int e = 0; e < EPS; ++e) v[i][e] = p[(long)e * ld];
        }
    }
    __device__ __forceinline__ void store(char* S) const {
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = threadIdx.x + NT * i;
            *reinterpret_cast<uint4*>(S + lds_off(idx % ROWS, idx / ROWS)) = pack_slot<F32>(v[i]);
        }
    }
};

// One LDS buffer, two barriers per K-tile: the HBM/L2 latency of the register-staged loads is
// hidden by thread-level parallelism (several workgroups per CU), which measured faster than
// double buffering at equal tile size.  NW = 4 waves (2x2) or 8 waves (2x4) per workgroup: with 8
// waves each thread stages half as much (fewer VGPRs -> more resident waves -> more loads in flight).
template <int BM, int BN, bool F32, bool A_KMAJ, bool B_KMAJ, int NW>
__global__ __launch_bounds__(NW * 64) void cst_gemm_kernel(GemmArgs g) {
    constexpr int BK = TileCfg<F32>::BK;
    constexpr int NT = NW * 64, NBUF = 1;
    constexpr int WGN = NW / 2;                    // waves along N (2 or 4); always 2 along M
    constexpr int TM = BM / 32, TN = BN / (16 * WGN);   // 16x16 tiles per wave in each direction
    constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[NBUF * (A_BYTES + B_BYTES)];

    // block -> tile.  Blocks are dealt round-robin over the 8 XCDs (speed only, never correctness):
    // give each XCD one contiguous run of tile ids, and order ids so that 4 consecutive ones share
    // an A row-panel while cycling through 4 B column-panels -- both stay resident in that XCD's L2.
    const int tilesM = (g.M + BM - 1) / BM, tilesN = (g.N + BN - 1) / BN;
    int id;
    {
        const int nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = blockIdx.x & 7;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
    }
    constexpr int GN = 4;
    const int grp = id / (GN * tilesM);
    const int gw = min(GN, tilesN - grp * GN);
    const int local = id - grp * GN * tilesM;
    const int tm = local / gw, tn = grp * GN + local % gw;
    const int m0 = tm * BM, n0 = tn * BN;
    const long bz = blockIdx.z;
    const float* A = g.A + bz * g.sA;
    const float* B = g.B + bz * g.sB;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int lr = lane & 15, lq = lane >> 4;

    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    Stage<BM, F32, A_KMAJ, NT> sa;
    Stage<BN, F32, B_KMAJ, NT> sb;
    const int kbeg = blockIdx.y * g.k_per_split;
    const int kend = min(g.K, kbeg + g.k_per_split);
    const int nk = (kend - kbeg + BK - 1) / BK;
    // wave-uniform choice per K-tile: the branch-free loader needs a full tile (and 16-byte
    // addressable rows for K-major operands); only the K tail takes the guarded loader
    const bool fastA = A_KMAJ ? g.vecA : true, fastB = B_KMAJ ? g.vecB : true;
    auto load_tiles = [&](int k0) {
        const bool full = k0 + BK <= kend;
        if (full && fastA) sa.load_full(A, g.lda, m0, k0, g.M); else sa.load(A, g.lda, m0, k0, g.M, kend, g.vecA);
        if (full && fastB) sb.load_full(B, g.ldb, n0, k0, g.N); else sb.load(B, g.ldb, n0, k0, g.N, kend, g.vecB);
    };
    load_tiles(kbeg);
    sa.store(smem);
    sb.store(smem + A_BYTES);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const char* As = smem;
        const char* Bs = As + A_BYTES;
        if (kt + 1 < nk) load_tiles(kbeg + (kt + 1) * BK);   // next tile: HBM -> registers while this one is multiplied
        // two k-steps per tile; fragment slot of lane group lq in step kk is kk*4 + lq
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const uint4*>(As + lds_off(wm * (BM / 2) + i * 16 + lr, kk * 4 + lq));
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const uint4*>(Bs + lds_off(wn * (BN / WGN) + j * 16 + lr, kk * 4 + lq));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (!F32) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[i]),
                                                                            __builtin_bit_cast(bf16x8_t, bfr[j]), acc[i][j], 0, 0, 0);
                    } else {
                        // k order inside the step is permuted identically for A and B: MFMA e consumes
                        // k = 16*kk + 4*lq + e from lane group lq -- each k exactly once.
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].x), __uint_as_float(bfr[j].x), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].y), __uint_as_float(bfr[j].y), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].z), __uint_as_float(bfr[j].z), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(af[i].w), __uint_as_float(bfr[j].w), acc[i][j], 0, 0, 0);
                    }
                }
        }
        __syncthreads();                       // everyone is done reading the buffer
        if (kt + 1 < nk) {
            sa.store(smem);
            sb.store(smem + A_BYTES);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout of the 16x16 MFMA: col = lane&15, row = 4*(lane>>4) + reg ----
    if (g.splits > 1) {
        float* slab = g.slab + ((bz * g.splits + blockIdx.y) * (long)g.M) * g.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (BN / WGN) + j * 16 + lr;
                if (n >= g.N) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * (BM / 2) + i * 16 + lq * 4 + r;
                    if (m < g.M) slab[(long)m * g.N + n] = acc[i][j][r];
                }
            }
        return;
    }
    float* C = g.C + bz * g.sC;
    const float* bias = g.bias ? g.bias + bz * g.sBias : nullptr;
    const float* addend = g.addend ? g.addend + bz * g.sAdd : nullptr;
    const float* aux = g.aux ? g.aux + bz * g.sAux : nullptr;
    const uint32_t dseed = g.drop.p > 0.f ? cst_drop_seed(g.drop) : 0u;
    // Stage this wave's (BM/2)x(BN/2) accumulator block through LDS (the K-loop's last barrier has
    // retired every tile read) so that each global store instruction writes whole 256-byte row
    // segments instead of 64-byte pieces of four different rows.
    constexpr int WM = BM / 2, WN = BN / WGN, CLD = WN;   // 2-way ds_write_b32 conflicts are free
    constexpr int PASSES = 2;                              // stage half the rows of the wave block at a time
    constexpr int PR = WM / PASSES;                        // rows of the wave block per pass
    static_assert(NW * PR * CLD * 4 <= NBUF * (A_BYTES + B_BYTES), "C staging must fit the tile buffers");
    float* Cs = reinterpret_cast<float*>(smem) + wave * PR * CLD;
    static_assert((PR * (WN / 4)) % 64 == 0, "C staging rows must divide over the wave");
    constexpr int C4 = WN / 4;                             // float4 chunks per row of the wave block
    const bool cvec = (g.ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0);
    // Vector path: the wave's block is inside the matrix and every stream it touches moves as float4.  All of a pass's loads (old C when
    // accumulating, addend, aux) are requested before its first store and the stores are issued back to back: loads and stores share
    // one in-order counter (vmcnt), so a load behind a store costs that store's whole round trip -- the element-wise loop below pays one
    // per element (the straight-through product d p += dx E^T, an accumulate, ran 16 of them in a row per wave).
    constexpr int IT = PR * C4 / 64;
    static_assert(64 % C4 == 0, "a lane keeps its four columns in every iteration");
    const bool fast = cvec && m0 + wm * WM + WM <= g.M && n0 + wn * WN + WN <= g.N &&
                      (!addend || ((g.ldadd % 4 == 0) && ((reinterpret_cast<uintptr_t>(addend) & 15) == 0))) &&
                      (g.act < 3 || ((g.ldaux % 4 == 0) && ((reinterpret_cast<uintptr_t>(aux) & 15) == 0))) &&
                      (!bias || ((reinterpret_cast<uintptr_t>(bias) & 15) == 0));
    float b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (fast && bias) {
        const float4 t = *reinterpret_cast<const float4*>(bias + n0 + wn * WN + (lane % C4) * 4);
        b4[0] = t.x; b4[1] = t.y; b4[2] = t.z; b4[3] = t.w;
    }
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
        if (ps > 0) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < TM / PASSES; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) Cs[(i * 16 + lq * 4 + r) * CLD + j * 16 + lr] = acc[ps * (TM / PASSES) + i][j][r];
        __builtin_amdgcn_wave_barrier();
        if (fast) {
            float4 a4[IT], old[IT], ad[IT], ax[IT];
            const int cc = (lane % C4) * 4, n = n0 + wn * WN + cc;
            const int mrow = m0 + wm * WM + ps * PR + lane / C4;
#pragma unroll
            for (int it = 0; it < IT; ++it) a4[it] = *reinterpret_cast<const float4*>(&Cs[(lane / C4 + (64 / C4) * it) * CLD + cc]);
            if (g.accumulate) {
#pragma unroll
                for (int it = 0; it < IT; ++it) old[it] = *reinterpret_cast<const float4*>(C + (long)(mrow + (64 / C4) * it) * g.ldc + n);
            }
            if (addend) {
#pragma unroll
                for (int it = 0; it < IT; ++it) ad[it] = *reinterpret_cast<const float4*>(addend + (long)(mrow + (64 / C4) * it) * g.ldadd + n);
            }
            if (g.act >= 3) {
#pragma unroll
                for (int it = 0; it < IT; ++it) ax[it] = *reinterpret_cast<const float4*>(aux + (long)(mrow + (64 / C4) * it) * g.ldaux + n);
            }
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                const int m = mrow + (64 / C4) * it;
                float o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = g.alpha * (&a4[it].x)[e] + b4[e];
                    if (addend) v += (&ad[it].x)[e];
                    if (g.act == 1) v = v > 0.f ? v : 0.f;
                    else if (g.act == 2) v = v > 0.f ? v : 0.1f * v;
                    else if (g.act == 3) v = (&ax[it].x)[e] > 0.f ? v * g.gate_scale : 0.f;
                    else if (g.act == 4) v = (&ax[it].x)[e] > 0.f ? v : 0.1f * v;
                    if (g.drop.p > 0.f) v *= cst_drop_mask(g.drop, dseed, (uint32_t)((long)m * g.N + n + e + bz * (long)g.M * g.N));
                    if (g.accumulate) v += (&old[it].x)[e];
                    o[e] = v;
                }
                *reinterpret_cast<float4*>(C + (long)m * g.ldc + n) = make_float4(o[0], o[1], o[2], o[3]);
            }
            continue;
        }
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int idx = lane + 64 * it;
            const int rr = idx / C4, cc = (idx % C4) * 4;
            const int m = m0 + wm * WM + ps * PR + rr, n = n0 + wn * WN + cc;
            if (m >= g.M || n >= g.N) continue;
            const float4 a4 = *reinterpret_cast<const float4*>(&Cs[rr * CLD + cc]);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < g.N) gemm_epilogue_store(g, C, bias, addend, aux, dseed, bz, m, n + e, av[e]);
        }
    }
}

// split-K second phase: sum the partial slabs in split order (deterministic) and apply the epilogue
__global__ __launch_bounds__(256) void cst_gemm_splitk_reduce(GemmArgs g) {
    const long bz = blockIdx.z;
    float* C = g.C + bz * g.sC;
    const float* bias = g.bias ? g.bias + bz * g.sBias : nullptr;
    const float* addend = g.addend ? g.addend + bz * g.sAdd : nullptr;
    const float* aux = g.aux ? g.aux + bz * g.sAux : nullptr;
    const uint32_t dseed = g.drop.p > 0.f ? cst_drop_seed(g.drop) : 0u;
    const long MN = (long)g.M * g.N;
    const float* slab = g.slab + bz * g.splits * MN;
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < MN; e += (long)gridDim.x * 256) {
        float acc = 0.f;
        for (int s = 0; s < g.splits; ++s) acc += slab[s * MN + e];
        gemm_epilogue_store(g, C, bias, addend, aux, dseed, bz, (int)(e / g.N), (int)(e % g.N), acc);
    }
}

// ---- optional kernel-precise timing (bench.py's roofline leg) ------------------------------------
// When enabled, every cst_gemm_kernel launch goes through hipExtLaunchKernelGGL with a start and a
// stop event bound to that kernel (begin / end of the dispatch itself, like a profiler's kernel
// trace), so the measured duration excludes host launch gaps.
#include <hip/hip_ext.h>
#include <vector>
struct GemmProf { hipEvent_t a, b; double flops; double bytes; int which; int m, n, k; };    // which: 0 cst_gemm_kernel, 1 cst_gemm_bf16_kernel, 2 cst_gemm_bf16_pp_kernel
static bool g_prof_on = false;
static std::vector<GemmProf> g_prof;
bool cst_prof_on() { return g_prof_on; }
void cst_prof_push(hipEvent_t a, hipEvent_t b, double flops, double bytes, int which) { g_prof.push_back({a, b, flops, bytes, which, 0, 0, 0}); }
void cst_prof_push_shape(hipEvent_t a, hipEvent_t b, double flops, double bytes, int which, int m, int n, int k) {
    g_prof.push_back({a, b, flops, bytes, which, m, n, k});
}

// per-launch records since enable: mnk[3 i .. 3 i + 2] = M, N, K as the kernel saw them (0 where a call site does not record
// them), ms[i] = kernel duration, which[i] as above.  Does not clear the list (cst_gemm_profile_read(1, ...) does).
extern "C" int cst_gemm_profile_shapes(long max_records, int* mnk, double* ms, int* which, long* count) {
    long n = 0;
    for (auto& p : g_prof) {
        if (n >= max_records) break;
        if (hipEventSynchronize(p.b) != hipSuccess) { cst_set_error("cst_gemm_profile_shapes: event sync failed"); return CST_ERR_LAUNCH; }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, p.a, p.b);
        mnk[3 * n] = p.m; mnk[3 * n + 1] = p.n; mnk[3 * n + 2] = p.k;
        ms[n] = t; which[n] = p.which;
        ++n;
    }
    *count = n;
    return CST_OK;
}

extern "C" int cst_gemm_profile_enable(int on) {
    if (on && !g_prof_on) g_prof.clear();
    g_prof_on = on != 0;
    return CST_OK;
}

// sums over the recorded launches of kernel `which` (0 = cst_gemm_kernel, 1 = cst_gemm_bf16_kernel, 2 = cst_gemm_bf16_pp_kernel);
// reading with which = 1 also destroys all events and clears the record list
extern "C" int cst_gemm_profile_read(int which, double* total_ms, double* total_flops, double* total_min_bytes, long* launches) {
    double ms = 0, fl = 0, by = 0;
    long n = 0;
    for (auto& p : g_prof) {
        if (p.which != which) continue;
        if (hipEventSynchronize(p.b) != hipSuccess) { cst_set_error("cst_gemm_profile_read: event sync failed"); return CST_ERR_LAUNCH; }
        float t = 0.f;
        (void)hipEventElapsedTime(&t, p.a, p.b);
        ms += t; fl += p.flops; by += p.bytes; ++n;
    }
    *total_ms = ms; *total_flops = fl; *total_min_bytes = by; *launches = n;
    if (which == 1) {
        for (auto& p : g_prof) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
        g_prof.clear();
    }
    return CST_OK;
}

template <int BM, int BN, int NW>
static void launch_cfg(const GemmArgs& g, int f32, int akm, int bkm, int batch, hipStream_t st) {
    dim3 grid(cst_div_up(g.M, BM) * cst_div_up(g.N, BN), g.splits, batch), block(NW * 64);
    hipEvent_t ea = nullptr, eb = nullptr;
    if (g_prof_on) {
        (void)hipEventCreate(&ea); (void)hipEventCreate(&eb);
        g_prof.push_back({ea, eb, 2.0 * g.M * g.N * g.K * batch, 4.0 * ((double)g.M * g.K + (double)g.N * g.K + (double)g.M * g.N) * batch, 0, g.M, g.N, g.K});
    }
#define CST_GEMM_CASE(F, AK, BKM)                                                              \
    if (f32 == F && akm == AK && bkm == BKM) {                                                 \
        if (g_prof_on) hipExtLaunchKernelGGL((cst_gemm_kernel<BM, BN, (bool)F, (bool)AK, (bool)BKM, NW>), grid, block, 0, st, ea, eb, 0, g); \
        else hipLaunchKernelGGL((cst_gemm_kernel<BM, BN, (bool)F, (bool)AK, (bool)BKM, NW>), grid, block, 0, st, g); \
        return;                                                                                \
    }
    CST_GEMM_CASE(0, 1, 1) CST_GEMM_CASE(0, 1, 0) CST_GEMM_CASE(0, 0, 1) CST_GEMM_CASE(0, 0, 0)
    CST_GEMM_CASE(1, 1, 1) CST_GEMM_CASE(1, 1, 0) CST_GEMM_CASE(1, 0, 1) CST_GEMM_CASE(1, 0, 0)
#undef CST_GEMM_CASE
}

static int vec_ok(const float* p, long ld, long stride) {
    return (((uintptr_t)p & 15) == 0) && (ld % 4 == 0) && (stride % 4 == 0);
}

// Tile / split-K choice of cst_gemm, shared with cst_gemm_workspace_floats so the two cannot drift: 128x128 tiles when they fill
// the 256 CUs on their own; with a long K and a handful of big tiles, big tiles + split-K; otherwise 64x64 tiles, split along K when
// even those are few.  Partial slabs are summed in slice order by cst_gemm_splitk_reduce.  The split count shrinks to what fits
// `ws_floats`.
//
// Exact mode (precision_f32): an output element is a k-ordered fma chain per K slice plus the slices in order, so its bits depend on
// the split count and on nothing else.  north_star asks for bit-exact greedy ids, also when the batch is sharded over data-parallel
// ranks: the split count of the exact mode is therefore chosen from (N, K, batch) for a NOMINAL 256-row product, never from M, and a
// product whose slabs do not fit the workspace is cut into row blocks (cst_gemm) instead of getting fewer, M-dependent slices.
constexpr int GEMM_EXACT_NOMINAL_M = 256;
static void gemm_plan(int M, int N, int K, int batch, int precision_f32, int tile, int splitk, bool has_ws, long ws_floats,
                      int* use_big_out, int* splits_out, int* kps_out, bool fit_rows = false) {
    const int Mp = precision_f32 ? GEMM_EXACT_NOMINAL_M : M;
    const long big = (long)cst_div_up(Mp, 128) * cst_div_up(N, 128) * batch;
    const long small = (long)cst_div_up(Mp, 64) * cst_div_up(N, 64) * batch;
    int use_big, splits = 1;
    tile &= ~1;
    if (tile == 128 || (tile == 0 && big >= 192)) { use_big = 1; }
    else if (tile == 0 && K >= 2048 && big >= 16 && has_ws && splitk == 0) {
        use_big = 1;
        splits = (int)((512 + big - 1) / big);
        if (splits > K / 512) splits = K / 512;
    } else {
        use_big = 0;
        if (splitk == 0 && has_ws && small < 128 && K >= 512) {
            splits = (int)((512 + small - 1) / small);
            if (splits > K / 128) splits = K / 128;
        }
    }
    if (splitk > 1) splits = splitk;
    if (splitk == 1) splits = 1;
    // the TILE may follow the real row count (it does not enter the summation order): big tiles once they fill the chip
    if (precision_f32 && (tile & ~1) == 0 && (long)cst_div_up(M, 128) * cst_div_up(N, 128) * batch >= 192) use_big = 1;
    int kps = cst_div_up(K, 64) * 64;
    if (splits > 1) {
        const int bk = precision_f32 ? 32 : 64;
        kps = cst_div_up(cst_div_up(K, splits), bk) * bk;
        splits = cst_div_up(K, kps);
        while (!fit_rows && splits > 1 && (long)batch * splits * M * N > ws_floats) {
            kps += bk;
            splits = cst_div_up(K, kps);
        }
    }
    if (splits <= 1) { splits = 1; kps = cst_div_up(K, 64) * 64; }
    *use_big_out = use_big; *splits_out = splits; *kps_out = kps;
}

extern "C" long cst_gemm_workspace_floats(int M, int N, int K, int batch, int precision_f32, int tile, int splitk) {
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return 0;
    int use_big, splits, kps;
    gemm_plan(M, N, K, batch, precision_f32, tile, splitk, true, 0x7fffffffffffffffL, &use_big, &splits, &kps);
    return splits > 1 ? (long)batch * splits * M * N : 0;
}

extern "C" int cst_gemm(const float* A, long lda, int a_kmajor, const float* B, long ldb, int b_kmajor,
                        float* C, long ldc, int M, int N, int K,
                        const float* bias, const float* addend, long ldadd,
                        const float* aux, long ldaux, int act, float gate_scale,
                        int accumulate, float alpha, int precision_f32,
                        int batch, long sA, long sB, long sC, long sBias, long sAdd, long sAux,
                        float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                        int tile, int splitk, float* workspace, long workspace_floats, void* stream) {
    CST_REQUIRE(A && B && C, "cst_gemm: null operand");
    CST_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "cst_gemm: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
    CST_REQUIRE(lda >= (a_kmajor ? K : M), "cst_gemm: lda=%ld too small", lda);
    CST_REQUIRE(ldb >= (b_kmajor ? K : N), "cst_gemm: ldb=%ld too small", ldb);
    CST_REQUIRE(ldc >= N, "cst_gemm: ldc=%ld < N=%d", ldc, N);
    CST_REQUIRE(act >= 0 && act <= 4, "cst_gemm: bad act %d", act);
    CST_REQUIRE(!(act >= 3) || aux, "cst_gemm: gate activation needs aux");
    CST_REQUIRE(!addend || ldadd >= N, "cst_gemm: ldadd too small");
    CST_REQUIRE(!aux || ldaux >= N, "cst_gemm: ldaux too small");
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.addend = addend; g.aux = aux;
    g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldadd = ldadd; g.ldaux = ldaux;
    g.sA = sA; g.sB = sB; g.sC = sC; g.sBias = sBias; g.sAdd = sAdd; g.sAux = sAux;
    g.M = M; g.N = N; g.K = K; g.act = act; g.accumulate = accumulate;
    g.vecA = vec_ok(A, lda, sA); g.vecB = vec_ok(B, ldb, sB);
    g.alpha = alpha; g.gate_scale = gate_scale;
    g.drop = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)batch * M * N);
    // tile / split choice.  128x128 tiles when they fill the 256 CUs on their own; with a long K
    // and a handful of big tiles, big tiles + split-K; otherwise 64x64 tiles, split along K when
    // even those are few.  Partial slabs are summed in slice order by cst_gemm_splitk_reduce.
    const int w8 = tile & 1;                    // odd tile code 129 forces the 8-wave build, 128 the 4-wave one
    const int forced = tile != 0;
    int use_big, splits;
    // exact mode, one problem, no dropout (its index is the row number inside the launch): row blocks instead of fewer K slices
    const bool fit_rows = precision_f32 && batch == 1 && drop_p <= 0.f;
    gemm_plan(M, N, K, batch, precision_f32, tile, splitk, workspace != nullptr, workspace_floats, &use_big, &splits, &g.k_per_split, fit_rows);
    CST_REQUIRE(splits == 1 || workspace, "cst_gemm: split-K needs a workspace");
    g.splits = splits;
    g.slab = workspace;
    hipStream_t st = (hipStream_t)stream;
    const int pf = precision_f32 ? 1 : 0, ak = a_kmajor ? 1 : 0, bk_ = b_kmajor ? 1 : 0;
    int rows_per_launch = M;
    if (splits > 1 && (long)batch * splits * M * N > workspace_floats) {          // only reachable with fit_rows
        rows_per_launch = (int)(workspace_floats / ((long)splits * N)) / 128 * 128;
        CST_REQUIRE(rows_per_launch >= 128, "cst_gemm: workspace of %ld floats cannot hold %d K slices of a 128-row block (N=%d)", workspace_floats, splits, N);
    }
    for (int m0 = 0; m0 < M; m0 += rows_per_launch) {
        GemmArgs h = g;
        h.M = M - m0 < rows_per_launch ? M - m0 : rows_per_launch;
        h.A = A + (a_kmajor ? (long)m0 * lda : (long)m0);
        h.C = C + (long)m0 * ldc;
        if (addend) h.addend = addend + (long)m0 * ldadd;
        if (aux) h.aux = aux + (long)m0 * ldaux;
        h.vecA = vec_ok(h.A, lda, sA);
        // measured (tools/gemm_bench.py): two K-major operands run best with 4 waves per workgroup, any
        // MN-major operand (dgrad / wgrad: many scalar staging loads) with 8
        if (use_big && (forced ? w8 : !(ak && bk_))) launch_cfg<128, 128, 8>(h, pf, ak, bk_, batch, st);
        else if (use_big) launch_cfg<128, 128, 4>(h, pf, ak, bk_, batch, st);
        else launch_cfg<64, 64, 4>(h, pf, ak, bk_, batch, st);
        CST_LAUNCH_CHECK("cst_gemm");
        if (splits > 1) {
            long mn = (long)h.M * N;
            int rb = (int)((mn + 255) / 256); if (rb > 2048) rb = 2048;
            hipLaunchKernelGGL(cst_gemm_splitk_reduce, dim3(rb, 1, batch), dim3(256), 0, st, h);
            CST_LAUNCH_CHECK("cst_gemm_splitk_reduce");
        }
    }
    return CST_OK;
}
