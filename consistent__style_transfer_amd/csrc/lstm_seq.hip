// Whole-sequence BiLSTM encoder recurrences for gfx950 (reference: the nn.LSTM(bidirectional=True) of
// src/model/rnn.py:25-27 called at rnn.py:57,62), one launch per pass instead of two launches per time step.
//
// A recurrence is sequential in time but independent across batch rows, so a workgroup takes 16 batch rows of one
// direction through ALL L' steps with no inter-workgroup synchronisation: B/16 x 2 workgroups, each a chain of
// L' steps of
//     gates[16, 4H] = h_{t-1}[16, H] W_hh^T  (+ the precomputed input projection, bias included)
// on v_mfma_f32_16x16x32_bf16.  Wave w owns hidden units [64w, 64w+64) of all four gates, so the cell update of a
// (row, unit) pair finds its i, f, g, o pre-activations in the same lane and register slot of four accumulator
// tiles: the cell runs in registers, c never leaves them, and h_t goes back to LDS in bf16 as the next step's A
// operand.  W_hh (512 KB in bf16 at H = 256) does not fit the LDS and is streamed from L2 every step: 128 16-byte
// loads per lane and step, requested a tile ahead of the MFMAs that consume them.
//
// The backward pass mirrors it: dgates_t (cell backward, in registers) goes to LDS in bf16 as the A operand of
// dh_{t-1} = dgates_t W_hh, whose accumulator tiles land in the register slots the cell backward of step t-1 reads.
//
// H = 256 (4 waves x 64 units); B % 16 == 0; every fp32 row pointer 16-byte aligned.
#include "cst_common.h"

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

__device__ __forceinline__ bf16_t sq_f2bf(float v) { __bf16 h = (__bf16)v; return __builtin_bit_cast(unsigned short, h); }
// Hardware exp2 / reciprocal forms (v_exp_f32, v_rcp_f32; ~1e-6 relative): a workgroup does 4096 cell updates per step on
// four SIMDs, and the libm expf / tanhf expansions alone took 10 us of each step.
__device__ __forceinline__ float sq_sigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float sq_tanh(float x) { return 2.f * __frcp_rn(1.f + __expf(-2.f * x)) - 1.f; }

constexpr int SQ_H = 256;              // hidden units per direction
constexpr int SQ_J = SQ_H / 64;        // 16-unit tiles per gate per wave
constexpr int SQ_KK = SQ_H / 32;       // 32-wide k steps of h W_hh^T
constexpr int SQ_HS = SQ_H + 8;        // LDS row stride of the h tile (bf16 elements): 16-byte aligned, rows 4 banks apart

struct LstmSeqDir {
    const bf16_t* whh;                 // W_hh in bf16 FRAGMENT order [wave 4][gate 4][tile 4][k step 8][lane 64][8]: see cst_lstm_seq_fwd
    const float* xp;                   // [B, L*4H] input projection incl. both biases
    const float* h0;                   // [B, .] initial hidden state (leading dimension ldh0)
    float* gates;                      // [L, B, 4H] activated gates (kept for backward)
    float* cenc;                       // [L, B, H]  cell states of steps 0..L-2 (slot of the step's time index)
    float* hprev;                      // [B, L*H]   h entering each time index (B operand of dW_hh)
    bf16_t* hprevb;                    // optional bf16 twin of hprev (operand of the transposed-read weight-gradient product)
    float* c_last;                     // [B, .]     cell state after the last step (leading dimension ldcl)
    int reverse;                       // 0: t = n, 1: t = L-1-n
};
struct LstmSeqArgs {
    LstmSeqDir dir[2];
    float* mem;                        // [B, L*2H]  h_t of direction d at column t*2H + d*H
    bf16_t* memb;                      // same layout, bf16
    int B, L;
    long ldw, ldh0, ldcl;
};

// W_hh tiles kept in LDS for the whole launch: the last SQ_NL of a wave's 16 fragment tiles (8 KiB each).  The recurrence is bound by
// the W_hh stream out of L2 (512 KB per step and workgroup at ~50 GB/s per CU against ~2 us of MFMA): every tile that stays on chip
// is one the step does not wait for.  4 waves x 4 tiles x 8 KiB = 128 KiB next to the 8 KiB h tile.
constexpr int SQ_NL = 4;
constexpr int SQ_FWD_LDS = 16 * SQ_HS * 2 + 4 * SQ_NL * SQ_KK * 1024;
// ... and the FIRST SQ_NR tiles stay in REGISTERS for the whole launch: the kernel runs one wave per SIMD (512 VGPRs a lane), of which the
// step itself needs ~380.  SQ_NR x 8 x 16 B per lane = another SQ_NR x 8 KiB per wave that is never streamed again (2 tiles: 506 registers, no
// spill; 3 spill 10, 4 spill 24): with 4 + 2 of the 16 tiles on chip the per-step W_hh stream out of L2 is 320 instead of 384 KB per workgroup.
constexpr int SQ_NR = 2;

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_seq_fwd_kernel(LstmSeqArgs a) {
    constexpr int H = SQ_H;
    extern __shared__ __attribute__((aligned(16))) char sq_smem[];
    bf16_t* hA = reinterpret_cast<bf16_t*>(sq_smem);                       // [16][SQ_HS]
    const int d = blockIdx.y;
    const LstmSeqDir& D = a.dir[d];
    const int r0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int L = a.L, B = a.B;
    // h_{-1}: bf16 into the A tile, fp32 into hprev at the first time index
    {
        const int t0 = D.reverse ? L - 1 : 0;
        for (int i = threadIdx.x; i < 16 * H; i += 256) {
            const int r = i / H, u = i - r * H;
            const float v = D.h0[(long)(r0 + r) * a.ldh0 + u];
            hA[r * SQ_HS + u] = sq_f2bf(v);
            D.hprev[(long)(r0 + r) * L * H + (long)t0 * H + u] = v;
            if (D.hprevb) D.hprevb[(long)(r0 + r) * L * H + (long)t0 * H + u] = sq_f2bf(v);
        }
    }
    float c[SQ_J][4];
#pragma unroll
    for (int j = 0; j < SQ_J; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[j][r] = 0.f;
    // B fragment of (gate q, tile j, k step kk) = W_hh row q*H + 64w + 16j + lr, columns 32kk + 8lq .. +7, stored so that the
    // 64 lanes of one load instruction read one contiguous KiB (row-scattered 16-byte pieces keep the texture path busy
    // 16 cache lines per instruction and ran this kernel at 1/6 of its speed)
    const bf16_t* wfrag = D.whh + ((long)w * 16 * SQ_KK * 64 + lane) * 8;
    // this wave's resident tiles: [tile 16 - SQ_NL ..][k step][lane][16 B], filled once
    char* wl = sq_smem + 16 * SQ_HS * 2 + (w * SQ_NL * SQ_KK * 64 + lane) * 16;
#pragma unroll
    for (int i = 0; i < SQ_NL * SQ_KK; ++i)
        *reinterpret_cast<u32x4_t*>(wl + i * 1024) = *reinterpret_cast<const u32x4_t*>(wfrag + ((long)(16 - SQ_NL) * SQ_KK + i) * 64 * 8);

    u32x4_t wreg[SQ_NR][SQ_KK];
#pragma unroll
    for (int ti = 0; ti < SQ_NR; ++ti)
#pragma unroll
        for (int kk = 0; kk < SQ_KK; ++kk) wreg[ti][kk] = *reinterpret_cast<const u32x4_t*>(wfrag + ((long)ti * SQ_KK + kk) * 64 * 8);

    for (int n = 0; n < L; ++n) {
        const int t = D.reverse ? L - 1 - n : n;
        f32x4_t acc[4][SQ_J];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < SQ_J; ++j) acc[q][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // W_hh fragments, three tiles in flight
        u32x4_t bb[3][SQ_KK];
        auto load_b = [&](int ti, u32x4_t (&dst)[SQ_KK]) {
            if (ti < SQ_NR) return;                       // register-resident tile: used in place below
            if (ti >= 16 - SQ_NL) {                       // resident tile (ti is a compile-time constant once the loop is unrolled)
                const char* ws = wl + (ti - (16 - SQ_NL)) * SQ_KK * 1024;
#pragma unroll
                for (int kk = 0; kk < SQ_KK; ++kk) dst[kk] = *reinterpret_cast<const u32x4_t*>(ws + kk * 1024);
                return;
            }
            const bf16_t* wn = wfrag + (long)ti * SQ_KK * 64 * 8;
#pragma unroll
            for (int kk = 0; kk < SQ_KK; ++kk) dst[kk] = *reinterpret_cast<const u32x4_t*>(wn + kk * 64 * 8);
        };
        load_b(0, bb[0]);
        load_b(1, bb[1]);
        __syncthreads();                                  // h tile of this step complete (and last step's reads done)
        u32x4_t af[SQ_KK];
#pragma unroll
        for (int kk = 0; kk < SQ_KK; ++kk)
            af[kk] = *reinterpret_cast<const u32x4_t*>(&hA[lr * SQ_HS + kk * 32 + lq * 8]);
        __syncthreads();                                  // every wave holds its A fragments: the tile may be overwritten
#pragma unroll
        for (int ti = 0; ti < 4 * SQ_J; ++ti) {
            if (ti + 2 < 4 * SQ_J) load_b(ti + 2, bb[(ti + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);            // keep two tiles of loads in flight: the scheduler otherwise sinks them next to their use
            f32x4_t s = acc[ti / SQ_J][ti % SQ_J];
#pragma unroll
            for (int kk = 0; kk < SQ_KK; ++kk)
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[kk]),
                                                            __builtin_bit_cast(bf16x8_t, ti < SQ_NR ? wreg[ti < SQ_NR ? ti : 0][kk] : bb[ti % 3][kk]), s, 0, 0, 0);
            acc[ti / SQ_J][ti % SQ_J] = s;
            __builtin_amdgcn_sched_barrier(0);
        }
        // input projection of this time index: requested only now -- the VMEM counter holds 63 outstanding operations, and
        // 64 more loads in flight across the tile loop would force it to drain between tiles
        float xv[4][SQ_J][4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < SQ_J; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xv[q][j][r] = D.xp[(long)(r0 + 4 * lq + r) * L * 4 * H + (long)t * 4 * H + q * H + 64 * w + 16 * j + lr];
        // cell: accumulator element r of tile (q, j) is gate q of (row 4lq + r, unit 64w + 16j + lr)
        const bool last = n == L - 1;
        const int tn = D.reverse ? t - 1 : t + 1;          // time index of the next step
#pragma unroll
        for (int j = 0; j < SQ_J; ++j) {
            const int u = 64 * w + 16 * j + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + 4 * lq + r;
                const float gi = sq_sigmoid(acc[0][j][r] + xv[0][j][r]);
                const float gf = sq_sigmoid(acc[1][j][r] + xv[1][j][r]);
                const float gg = sq_tanh(acc[2][j][r] + xv[2][j][r]);
                const float go = sq_sigmoid(acc[3][j][r] + xv[3][j][r]);
                const float cn = gf * c[j][r] + gi * gg;
                const float h = go * sq_tanh(cn);
                c[j][r] = cn;
                float* g = D.gates + ((long)t * B + row) * 4 * H + u;
                g[0] = gi; g[H] = gf; g[2 * H] = gg; g[3 * H] = go;
                if (last) D.c_last[(long)row * a.ldcl + u] = cn;
                else D.cenc[((long)t * B + row) * H + u] = cn;
                const long mo = (long)row * L * 2 * H + (long)t * 2 * H + d * H + u;
                a.mem[mo] = h;
                const bf16_t hb = sq_f2bf(h);
                a.memb[mo] = hb;
                if (!last) {
                    D.hprev[(long)row * L * H + (long)tn * H + u] = h;
                    if (D.hprevb) D.hprevb[(long)row * L * H + (long)tn * H + u] = hb;
                }
                hA[(4 * lq + r) * SQ_HS + u] = hb;
            }
        }
    }
}

extern "C" int cst_lstm_seq_fwd(const void* whh0, const void* whh1, const float* xp0, const float* xp1,
                                const float* h0, long ldh0, float* gates0, float* gates1, float* cenc0, float* cenc1,
                                float* hprev0, float* hprev1, void* hprev0_bf16, void* hprev1_bf16,
                                float* c_last, long ldcl, float* mem, void* mem_bf16,
                                int B, int L, int H, void* stream) {
    CST_REQUIRE(whh0 && whh1 && xp0 && xp1 && h0 && gates0 && gates1 && cenc0 && cenc1 && hprev0 && hprev1 && c_last && mem && mem_bf16,
                "cst_lstm_seq_fwd: null pointer");
    CST_REQUIRE(H == SQ_H && B > 0 && B % 16 == 0 && L > 0, "cst_lstm_seq_fwd: needs H == %d and B %% 16 == 0 (H=%d, B=%d)", SQ_H, H, B);
    CST_REQUIRE(((((uintptr_t)whh0) | ((uintptr_t)whh1)) & 15) == 0, "cst_lstm_seq_fwd: W_hh fragment copies must be 16-byte aligned");
    LstmSeqArgs a;
    CST_REQUIRE(!hprev0_bf16 == !hprev1_bf16, "cst_lstm_seq_fwd: pass both bf16 hprev twins or neither");
    a.dir[0] = LstmSeqDir{(const bf16_t*)whh0, xp0, h0, gates0, cenc0, hprev0, (bf16_t*)hprev0_bf16, c_last, 0};
    a.dir[1] = LstmSeqDir{(const bf16_t*)whh1, xp1, h0 + H, gates1, cenc1, hprev1, (bf16_t*)hprev1_bf16, c_last + H, 1};
    a.mem = mem; a.memb = (bf16_t*)mem_bf16; a.B = B; a.L = L; a.ldw = 0; a.ldh0 = ldh0; a.ldcl = ldcl;
    static CstPerDevice attr_done;
    if (cst_first_on_device(attr_done)) {
        (void)hipFuncSetAttribute((const void*)lstm_seq_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SQ_FWD_LDS);
    }
    hipLaunchKernelGGL(lstm_seq_fwd_kernel, dim3(B / 16, 2), dim3(256), SQ_FWD_LDS, (hipStream_t)stream, a);
    CST_LAUNCH_CHECK("cst_lstm_seq_fwd");
    return CST_OK;
}

// =============================================================================================
// Split forward (round 3): TWO workgroups per (direction, 16-row group), each owning half of the hidden units of all four gates.
//
// The one-workgroup kernel above is bound by the W_hh stream out of L2 (320 of 512 KB per step and workgroup, ~8.7 us a step against
// ~1 us of MFMA).  Halved, a workgroup's share of W_hh is 256 KB and stays ON CHIP for the whole launch: 128 KB in LDS (tiles 4..7 of each
// wave) and 128 KB in registers (tiles 0..3: 128 VGPRs a lane) -- nothing is streamed, and the price is one exchange per step: each
// workgroup needs the other's half of h_t (16 rows x 128 units, 4 KB in bf16) before step t + 1.  That is the guide's smallest hand-off
// (cdna_hip_programming.md Guideline 16, recipe R2; MI355X_MICROARCH.md price list, handoff-1to1: ~1.0-1.5 us for <= 4 KB): the data IS the
// flag -- 8-byte granules {epoch = step + 1, two bf16 values} written with agent-scope (write-through) atomic stores and re-read with
// agent-scope atomic loads until every tag carries the epoch; two buffers by step parity (a workgroup can run at most one step ahead of its
// partner); the buffers are zeroed by a kernel in front of every launch (epochs repeat from launch to launch); every spin is bounded and
// reports through a timeout word instead of hanging.  The fragment order of W_hh is the one-workgroup kernel's: wave w of half p reads the
// fragments of old wave 2p + w / 2, tiles 2 (w % 2) + {0, 1}.
// =============================================================================================
constexpr int SQ2_NREG = 4;                                  // of a wave's 8 tiles (4 gates x 2): tiles 0..3 in registers, 4..7 in LDS
constexpr int SQ2_FWD_LDS = 16 * SQ_HS * 2 + 4 * (8 - SQ2_NREG) * SQ_KK * 1024;
constexpr int SQ2_GRAN = 1024;                               // granules per half and step
constexpr unsigned SQ2_SPIN_MAX = 1u << 17;                  // sweeps before a wave gives up (a normal exchange takes 1-10; this is ~0.2 s)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_seq_fwd2_kernel(LstmSeqArgs a, unsigned long long* xchg,
                                                                                                       unsigned* tmo) {
    constexpr int H = SQ_H;
    extern __shared__ __attribute__((aligned(16))) char sq_smem[];
    bf16_t* hA = reinterpret_cast<bf16_t*>(sq_smem);                       // [16][SQ_HS], all 256 units
    const int d = blockIdx.y, p = blockIdx.z;
    const LstmSeqDir& D = a.dir[d];
    const int r0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int L = a.L, B = a.B;
    const int ub = 128 * p + 32 * w;                                       // first hidden unit of this wave
    {
        const int t0 = D.reverse ? L - 1 : 0;
        for (int i = threadIdx.x; i < 16 * H; i += 256) {
            const int r = i / H, u = i - r * H;
            const float v = D.h0[(long)(r0 + r) * a.ldh0 + u];
            hA[r * SQ_HS + u] = sq_f2bf(v);
            if ((u >> 7) == p) {                                           // each half records its own units
                D.hprev[(long)(r0 + r) * L * H + (long)t0 * H + u] = v;
                if (D.hprevb) D.hprevb[(long)(r0 + r) * L * H + (long)t0 * H + u] = sq_f2bf(v);
            }
        }
    }
    float c[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[j][r] = 0.f;
    // tile ti = 2 q + j (gate q, units ub + 16 j ..) = old fragment tile (wave 2p + w / 2, gate q, tile 2 (w % 2) + j)
    const bf16_t* wfrag = D.whh + ((long)(2 * p + (w >> 1)) * 16 * SQ_KK * 64 + lane) * 8;
    auto frag_of = [&](int ti, int kk) { return wfrag + ((long)((ti >> 1) * 4 + 2 * (w & 1) + (ti & 1)) * SQ_KK + kk) * 64 * 8; };
    char* wl = sq_smem + 16 * SQ_HS * 2 + (w * (8 - SQ2_NREG) * SQ_KK * 64 + lane) * 16;
#pragma unroll
    for (int ti = SQ2_NREG; ti < 8; ++ti)
#pragma unroll
        for (int kk = 0; kk < SQ_KK; ++kk)
            *reinterpret_cast<u32x4_t*>(wl + ((ti - SQ2_NREG) * SQ_KK + kk) * 1024) = *reinterpret_cast<const u32x4_t*>(frag_of(ti, kk));
    u32x4_t wreg[SQ2_NREG][SQ_KK];
#pragma unroll
    for (int ti = 0; ti < SQ2_NREG; ++ti)
#pragma unroll
        for (int kk = 0; kk < SQ_KK; ++kk) wreg[ti][kk] = *reinterpret_cast<const u32x4_t*>(frag_of(ti, kk));
    // exchange buffers of this (direction, row group): [half][parity][SQ2_GRAN]; thread (w, lane) owns granules 4 (64 w + lane) + k
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    const long gbase = ((long)d * gridDim.x + blockIdx.x) * 2;
    gu64* mine = (gu64*)(xchg + ((gbase + p) * 2) * SQ2_GRAN + 4 * (64 * w + lane));
    gu64* theirs = (gu64*)(xchg + ((gbase + (p ^ 1)) * 2) * SQ2_GRAN + 4 * (64 * w + lane));
    const int upb = 128 * (p ^ 1) + 32 * w;                                // the partner thread (w, lane) holds units upb + 16 j + lr

    // input projection of a time index: 32 scattered 4-byte loads a lane, requested a step ahead (right after the MFMAs of the step before,
    // so that their latency runs next to the exchange's)
    float xv[4][2][4];
    auto load_x = [&](int t) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    xv[q][j][r] = D.xp[(long)(r0 + 4 * lq + r) * L * 4 * H + (long)t * 4 * H + q * H + ub + 16 * j + lr];
    };
    load_x(D.reverse ? L - 1 : 0);
    for (int n = 0; n < L; ++n) {
        const int t = D.reverse ? L - 1 - n : n;
        __syncthreads();                                  // h tile of this step complete (own half + the partner's)
        u32x4_t af[SQ_KK];
#pragma unroll
        for (int kk = 0; kk < SQ_KK; ++kk)
            af[kk] = *reinterpret_cast<const u32x4_t*>(&hA[lr * SQ_HS + kk * 32 + lq * 8]);
        __syncthreads();                                  // every wave holds its A fragments: the tile may be overwritten
        f32x4_t acc[8];
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
            f32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < SQ_KK; ++kk) {
                const u32x4_t bfr = ti < SQ2_NREG ? wreg[ti < SQ2_NREG ? ti : 0][kk]
                                                  : *reinterpret_cast<const u32x4_t*>(wl + ((ti - SQ2_NREG) * SQ_KK + kk) * 1024);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, af[kk]), __builtin_bit_cast(bf16x8_t, bfr), s, 0, 0, 0);
            }
            acc[ti] = s;
        }
        const bool last = n == L - 1;
        const int tn = D.reverse ? t - 1 : t + 1;
        // cell in registers; the step's results are kept until the exchange is under way
        float og[4][2][4], ocn[2][4], oh[2][4];
        unsigned hv[2][2];                                 // this lane's h values as bf16 pairs: [j][rows (0, 1) | rows (2, 3)]
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int u = ub + 16 * j + lr;
            bf16_t hb4[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gi = sq_sigmoid(acc[0 + j][r] + xv[0][j][r]);
                const float gf = sq_sigmoid(acc[2 + j][r] + xv[1][j][r]);
                const float gg = sq_tanh(acc[4 + j][r] + xv[2][j][r]);
                const float go = sq_sigmoid(acc[6 + j][r] + xv[3][j][r]);
                const float cn = gf * c[j][r] + gi * gg;
                const float h = go * sq_tanh(cn);
                c[j][r] = cn;
                og[0][j][r] = gi; og[1][j][r] = gf; og[2][j][r] = gg; og[3][j][r] = go;
                ocn[j][r] = cn; oh[j][r] = h;
                hb4[r] = sq_f2bf(h);
                hA[(4 * lq + r) * SQ_HS + u] = hb4[r];
            }
            hv[j][0] = (unsigned)hb4[0] | ((unsigned)hb4[1] << 16);
            hv[j][1] = (unsigned)hb4[2] | ((unsigned)hb4[3] << 16);
        }
        unsigned pv[4] = {0u, 0u, 0u, 0u};
        if (!last) {
            load_x(tn);                                    // next step's input projection: in flight during the exchange
            // publish this half of h_t, then take the partner's: granule k = 2 j + (row pair) of thread (w, lane).  The polling loads sit
            // behind the x loads in this wave's (in-order) memory queue, and in front of the step's output stores.
            const unsigned long long tag = (unsigned long long)(n + 1) << 32;
            gu64* mo_ = mine + (long)(n & 1) * SQ2_GRAN;
            gu64* to_ = theirs + (long)(n & 1) * SQ2_GRAN;
#pragma unroll
            for (int k = 0; k < 4; ++k) __hip_atomic_store(mo_ + k, tag | hv[k >> 1][k & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned spins = 0;; ++spins) {
                bool ok = true;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned long long x = __hip_atomic_load(to_ + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pv[k] = (unsigned)x;
                    ok &= (x >> 32) == (unsigned long long)(n + 1);
                }
                if (__all(ok)) break;
                if (spins >= SQ2_SPIN_MAX) {
                    // never hang -- and never compute on stale hidden states either: report through the sticky word and poison the
                    // partner's half of h_t with NaN, so that every later gate, mem row and c_last of this row group is NaN and the
                    // loss of THIS step fails loudly (the host reads the word at its next synchronisation point)
                    if (lane == 0) atomicOr(tmo, 1u);
#pragma unroll
                    for (int k = 0; k < 4; ++k) pv[k] = 0x7FC07FC0u;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int u = upb + 16 * (k >> 1) + lr, row = 4 * lq + 2 * (k & 1);
                hA[row * SQ_HS + u] = (bf16_t)(pv[k] & 0xffffu);
                hA[(row + 1) * SQ_HS + u] = (bf16_t)(pv[k] >> 16);
            }
        }
        // the step's outputs: fire and forget (they drain under the next step's barrier and MFMAs)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int u = ub + 16 * j + lr;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = r0 + 4 * lq + r;
                float* g = D.gates + ((long)t * B + row) * 4 * H + u;
                g[0] = og[0][j][r]; g[H] = og[1][j][r]; g[2 * H] = og[2][j][r]; g[3 * H] = og[3][j][r];
                if (last) D.c_last[(long)row * a.ldcl + u] = ocn[j][r];
                else D.cenc[((long)t * B + row) * H + u] = ocn[j][r];
                const long mo = (long)row * L * 2 * H + (long)t * 2 * H + d * H + u;
                a.mem[mo] = oh[j][r];
                const bf16_t hb = sq_f2bf(oh[j][r]);
                a.memb[mo] = hb;
                if (!last) {
                    D.hprev[(long)row * L * H + (long)tn * H + u] = oh[j][r];
                    if (D.hprevb) D.hprevb[(long)row * L * H + (long)tn * H + u] = hb;
                }
            }
        }
    }
}

// bytes of exchange workspace cst_lstm_seq_fwd_split needs for batch B (two directions x B / 16 row groups x 2 halves x 2 parities)
extern "C" long cst_lstm_seq_xchg_bytes(int B) { return B > 0 ? (long)2 * (B / 16) * 2 * 2 * SQ2_GRAN * 8 + 16 : 0; }

// CUs of the current device (one lookup per device; the split kernel's workgroups need a whole CU each: 136 KB of LDS, 512 registers a lane)
static int sq_device_cus() {
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus[dev] = n > 0 ? n : -1;
    }
    return cus[dev] > 0 ? cus[dev] : 0;
}

// workgroups cst_lstm_seq_fwd_split launches for batch B, and how many this device can keep resident together (its CU count): the caller
// (gen_fn.split_enabled) takes the split kernel only when the first is well inside the second
extern "C" int cst_lstm_seq_split_workgroups(int B) { return B > 0 ? (B / 16) * 4 : 0; }
extern "C" int cst_lstm_seq_split_capacity() { return sq_device_cus(); }

static int sq_fwd_split_launch(const char* who, int halves, const void* whh0, const void* whh1, const float* xp0, const float* xp1,
                               const float* h0, long ldh0, float* gates0, float* gates1, float* cenc0, float* cenc1,
                               float* hprev0, float* hprev1, void* hprev0_bf16, void* hprev1_bf16,
                               float* c_last, long ldcl, float* mem, void* mem_bf16,
                               int B, int L, int H, void* xchg, long xchg_bytes, void* stream) {
    CST_REQUIRE(whh0 && whh1 && xp0 && xp1 && h0 && gates0 && gates1 && cenc0 && cenc1 && hprev0 && hprev1 && c_last && mem && mem_bf16 && xchg,
                "%s: null pointer", who);
    CST_REQUIRE(H == SQ_H && B > 0 && B % 16 == 0 && L > 0 && B <= 1024, "%s: needs H == %d, B %% 16 == 0, B <= 1024 (H=%d, B=%d)", who, SQ_H, H, B);
    CST_REQUIRE(((((uintptr_t)whh0) | ((uintptr_t)whh1) | ((uintptr_t)xchg)) & 15) == 0, "%s: W_hh fragment copies / workspace must be 16-byte aligned", who);
    CST_REQUIRE(xchg_bytes >= cst_lstm_seq_xchg_bytes(B), "%s: exchange workspace of %ld bytes, need %ld", who, xchg_bytes, cst_lstm_seq_xchg_bytes(B));
    CST_REQUIRE(!hprev0_bf16 == !hprev1_bf16, "%s: pass both bf16 hprev twins or neither", who);
    // all 4 B / 16 workgroups must be resident together (a workgroup spins on its partner), one per CU: checked against THIS device's CU
    // count before anything is launched or captured.  Residency beside other kernels (RCCL, forked streams) is the caller's margin.
    const int cus = sq_device_cus();
    CST_REQUIRE(cus > 0 && (B / 16) * 4 <= cus, "%s: %d workgroups would not be co-resident on %d CUs", who, (B / 16) * 4, cus);
    LstmSeqArgs a;
    a.dir[0] = LstmSeqDir{(const bf16_t*)whh0, xp0, h0, gates0, cenc0, hprev0, (bf16_t*)hprev0_bf16, c_last, 0};
    a.dir[1] = LstmSeqDir{(const bf16_t*)whh1, xp1, h0 + H, gates1, cenc1, hprev1, (bf16_t*)hprev1_bf16, c_last + H, 1};
    a.mem = mem; a.memb = (bf16_t*)mem_bf16; a.B = B; a.L = L; a.ldw = 0; a.ldh0 = ldh0; a.ldcl = ldcl;
    hipStream_t st = (hipStream_t)stream;
    // the dynamic-LDS limit is a per-device property of the function: set it on every call (cheap), not once per process
    if (hipFuncSetAttribute((const void*)lstm_seq_fwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SQ2_FWD_LDS) != hipSuccess) {
        (void)hipGetLastError();
        cst_set_error("%s: cannot raise the dynamic LDS limit to %d bytes", who, SQ2_FWD_LDS);
        return CST_ERR_LAUNCH;
    }
    // every polled word starts at zero in every launch (epochs repeat); the last 16 bytes are the timeout word, which is sticky (zeroed
    // by whoever allocates the workspace, never here: a later launch must not erase an earlier one's report)
    const long bytes = cst_lstm_seq_xchg_bytes(B);
    if (cst_zero_words(xchg, (bytes - 16) / 4, st) != CST_OK) { cst_set_error("%s: zero fill failed", who); return CST_ERR_LAUNCH; }
    hipLaunchKernelGGL(lstm_seq_fwd2_kernel, dim3(B / 16, 2, halves), dim3(256), SQ2_FWD_LDS, st, a,
                       (unsigned long long*)xchg, (unsigned*)((char*)xchg + bytes - 16));
    CST_LAUNCH_CHECK(who);
    return CST_OK;
}

extern "C" int cst_lstm_seq_fwd_split(const void* whh0, const void* whh1, const float* xp0, const float* xp1,
                                      const float* h0, long ldh0, float* gates0, float* gates1, float* cenc0, float* cenc1,
                                      float* hprev0, float* hprev1, void* hprev0_bf16, void* hprev1_bf16,
                                      float* c_last, long ldcl, float* mem, void* mem_bf16,
                                      int B, int L, int H, void* xchg, long xchg_bytes, void* stream) {
    return sq_fwd_split_launch("cst_lstm_seq_fwd_split", 2, whh0, whh1, xp0, xp1, h0, ldh0, gates0, gates1, cenc0, cenc1, hprev0, hprev1,
                               hprev0_bf16, hprev1_bf16, c_last, ldcl, mem, mem_bf16, B, L, H, xchg, xchg_bytes, stream);
}

// TEST ONLY (tests/test_gpu_ops.py): the same launch with the second half of every workgroup pair missing, i.e. what a lost co-residency
// looks like -- every first-half workgroup runs into its bounded spin at the first exchange, sets the sticky word and poisons its outputs.
extern "C" int cst_lstm_seq_fwd_split_lone_half(const void* whh0, const void* whh1, const float* xp0, const float* xp1,
                                                const float* h0, long ldh0, float* gates0, float* gates1, float* cenc0, float* cenc1,
                                                float* hprev0, float* hprev1, void* hprev0_bf16, void* hprev1_bf16,
                                                float* c_last, long ldcl, float* mem, void* mem_bf16,
                                                int B, int L, int H, void* xchg, long xchg_bytes, void* stream) {
    return sq_fwd_split_launch("cst_lstm_seq_fwd_split_lone_half", 1, whh0, whh1, xp0, xp1, h0, ldh0, gates0, gates1, cenc0, cenc1, hprev0, hprev1,
                               hprev0_bf16, hprev1_bf16, c_last, ldcl, mem, mem_bf16, B, L, H, xchg, xchg_bytes, stream);
}

// =============================================================================================
// Backward of the same recurrences: per step the LSTM cell backward in registers, dgates (bf16) to LDS as the A
// operand of dh_{t-1}[16, H] = dgates[16, 4H] W_hh, whose accumulator tiles are the register slots the next cell
// backward reads.  W_hh^T is streamed in fragment order [wave][k step 32][tile 4][lane 64][8].
// =============================================================================================
constexpr int SQ_GS = 4 * SQ_H + 8;    // LDS row stride of the dgates tile (bf16 elements)
constexpr int SQ_KB = 4 * SQ_H / 32;   // 32-wide k steps over the 4H gate columns

struct LstmSeqBwdDir {
    const bf16_t* wt;                  // W_hh^T in bf16 fragment order (see cst_lstm_seq_bwd)
    const float* gates;                // [L, B, 4H] activated gates of the forward pass
    const float* cenc;                 // [L, B, H]
    const float* c_last;               // [B, .]  (ldcl)
    const float* dc_last;              // [B, .]  gradient w.r.t. the final cell state (lddcl)
    float* dgates;                     // [B, L*4H] pre-activation gate gradients (operand of the weight gradients)
    bf16_t* dgatesb;                   // optional bf16 twin of dgates
    float* dh0;                        // [B, .]  gradient w.r.t. the initial hidden state (lddh0)
    int reverse;
};
struct LstmSeqBwdArgs {
    LstmSeqBwdDir dir[2];
    const float* dmem;                 // [B, L*2H] gradient w.r.t. the encoder states (direction d at column t*2H + d*H)
    int B, L;
    long ldcl, lddcl, lddh0;
};

// resident W_hh^T fragments: the last SQ_NLB of a wave's 32 k steps (4 KiB each): 4 waves x 7 x 4 KiB = 112 KiB next to the 32 KiB dgates tile
constexpr int SQ_NLB = 7;
constexpr int SQ_BWD_LDS = 16 * SQ_GS * 2 + 4 * SQ_NLB * SQ_J * 1024;
constexpr int SQ_NRB = 16;             // ... and the FIRST SQ_NRB k steps stay in registers (one wave per SIMD: 512 VGPRs a lane): 9 of 32 k steps streamed per step instead of 25 (504 registers, no spill)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void lstm_seq_bwd_kernel(LstmSeqBwdArgs a) {
    constexpr int H = SQ_H;
    extern __shared__ __attribute__((aligned(16))) char sq_smem[];
    bf16_t* gA = reinterpret_cast<bf16_t*>(sq_smem);                       // [16][SQ_GS]
    const int d = blockIdx.y;
    const LstmSeqBwdDir& D = a.dir[d];
    const int r0 = blockIdx.x * 16;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane & 15, lq = lane >> 4;
    const int L = a.L, B = a.B;
    // The MFMAs below take W_hh^T as the FIRST operand: the 16 x 16 result comes out transposed, so a lane holds FOUR CONSECUTIVE units
    // (64 w + 16 j + 4 lq + r) of ONE batch row (lr).  Every operand of the cell backward is then a 16-byte load and every result a 16-byte
    // (fp32) or 8-byte (bf16) store: 28 loads and 32 stores a lane and step where the element-per-lane layout needed 112 and 128.
    const int row = r0 + lr;
    float dc[SQ_J][4], dhr[SQ_J][4];
#pragma unroll
    for (int j = 0; j < SQ_J; ++j) {
        const f32x4_t v = *reinterpret_cast<const f32x4_t*>(D.dc_last + (long)row * a.lddcl + 64 * w + 16 * j + 4 * lq);
#pragma unroll
        for (int r = 0; r < 4; ++r) { dc[j][r] = v[r]; dhr[j][r] = 0.f; }
    }
    const bf16_t* wfrag = D.wt + ((long)w * SQ_KB * SQ_J * 64 + lane) * 8;
    char* wl = sq_smem + 16 * SQ_GS * 2 + (w * SQ_NLB * SQ_J * 64 + lane) * 16;
#pragma unroll
    for (int i = 0; i < SQ_NLB * SQ_J; ++i)
        *reinterpret_cast<u32x4_t*>(wl + i * 1024) = *reinterpret_cast<const u32x4_t*>(wfrag + ((long)(SQ_KB - SQ_NLB) * SQ_J + i) * 64 * 8);

    u32x4_t wreg[SQ_NRB][SQ_J];
#pragma unroll
    for (int kk = 0; kk < SQ_NRB; ++kk)
#pragma unroll
        for (int j = 0; j < SQ_J; ++j) wreg[kk][j] = *reinterpret_cast<const u32x4_t*>(wfrag + ((long)kk * SQ_J + j) * 64 * 8);

    for (int n = L - 1; n >= 0; --n) {
        const int t = D.reverse ? L - 1 - n : n;
        const int tp = D.reverse ? t + 1 : t - 1;          // time index of the forward step before this one
        // ---- cell backward of step n (element r of tile j = (row lr, unit 64w + 16j + 4lq + r)) ----
#pragma unroll
        for (int j = 0; j < SQ_J; ++j) {
            const int u = 64 * w + 16 * j + 4 * lq;
            const float* g = D.gates + ((long)t * B + row) * 4 * H + u;
            const f32x4_t gi = *reinterpret_cast<const f32x4_t*>(g), gf = *reinterpret_cast<const f32x4_t*>(g + H);
            const f32x4_t gg = *reinterpret_cast<const f32x4_t*>(g + 2 * H), go = *reinterpret_cast<const f32x4_t*>(g + 3 * H);
            const f32x4_t cn = *reinterpret_cast<const f32x4_t*>((n == L - 1) ? D.c_last + (long)row * a.ldcl + u : D.cenc + ((long)t * B + row) * H + u);
            const f32x4_t cp = (n == 0) ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4_t*>(D.cenc + ((long)tp * B + row) * H + u);
            const f32x4_t dm = *reinterpret_cast<const f32x4_t*>(a.dmem + (long)row * L * 2 * H + (long)t * 2 * H + d * H + u);
            f32x4_t d0, d1, d2, d3;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dht = dm[r] + dhr[j][r];
                const float tc = sq_tanh(cn[r]);
                const float dct = dc[j][r] + dht * go[r] * (1.f - tc * tc);
                d0[r] = dct * gg[r] * gi[r] * (1.f - gi[r]);
                d1[r] = dct * cp[r] * gf[r] * (1.f - gf[r]);
                d2[r] = dct * gi[r] * (1.f - gg[r] * gg[r]);
                d3[r] = dht * tc * go[r] * (1.f - go[r]);
                dc[j][r] = dct * gf[r];
            }
            float* dg = D.dgates + (long)row * L * 4 * H + (long)t * 4 * H + u;
            *reinterpret_cast<f32x4_t*>(dg) = d0; *reinterpret_cast<f32x4_t*>(dg + H) = d1;
            *reinterpret_cast<f32x4_t*>(dg + 2 * H) = d2; *reinterpret_cast<f32x4_t*>(dg + 3 * H) = d3;
            auto pack4 = [](const f32x4_t& v) {
                uint2 q;
                q.x = (uint32_t)sq_f2bf(v[0]) | ((uint32_t)sq_f2bf(v[1]) << 16);
                q.y = (uint32_t)sq_f2bf(v[2]) | ((uint32_t)sq_f2bf(v[3]) << 16);
                return q;
            };
            const uint2 b0 = pack4(d0), b1 = pack4(d1), b2 = pack4(d2), b3 = pack4(d3);
            bf16_t* ga = gA + lr * SQ_GS + u;
            *reinterpret_cast<uint2*>(ga) = b0; *reinterpret_cast<uint2*>(ga + H) = b1;
            *reinterpret_cast<uint2*>(ga + 2 * H) = b2; *reinterpret_cast<uint2*>(ga + 3 * H) = b3;
            if (D.dgatesb) {
                bf16_t* gb = D.dgatesb + (long)row * L * 4 * H + (long)t * 4 * H + u;
                *reinterpret_cast<uint2*>(gb) = b0; *reinterpret_cast<uint2*>(gb + H) = b1;
                *reinterpret_cast<uint2*>(gb + 2 * H) = b2; *reinterpret_cast<uint2*>(gb + 3 * H) = b3;
            }
        }
        __syncthreads();                                  // the dgates tile is complete
        // ---- dh_{prev}[16, 64 units of this wave] = dgates[16, 4H] W_hh[4H, units] ----
        f32x4_t acc[SQ_J];
#pragma unroll
        for (int j = 0; j < SQ_J; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        u32x4_t bb[3][SQ_J];
        auto load_b = [&](int kk, u32x4_t (&dst)[SQ_J]) {
            if (kk < SQ_NRB) return;                      // register-resident k step: used in place below
            if (kk >= SQ_KB - SQ_NLB) {                   // resident k step
                const char* ws = wl + (kk - (SQ_KB - SQ_NLB)) * SQ_J * 1024;
#pragma unroll
                for (int j = 0; j < SQ_J; ++j) dst[j] = *reinterpret_cast<const u32x4_t*>(ws + j * 1024);
                return;
            }
            const bf16_t* wn = wfrag + (long)kk * SQ_J * 64 * 8;
#pragma unroll
            for (int j = 0; j < SQ_J; ++j) dst[j] = *reinterpret_cast<const u32x4_t*>(wn + j * 64 * 8);
        };
        load_b(0, bb[0]);
        load_b(1, bb[1]);
#pragma unroll
        for (int kk = 0; kk < SQ_KB; ++kk) {
            if (kk + 2 < SQ_KB) load_b(kk + 2, bb[(kk + 2) % 3]);
            const u32x4_t af = *reinterpret_cast<const u32x4_t*>(&gA[lr * SQ_GS + kk * 32 + lq * 8]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < SQ_J; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, kk < SQ_NRB ? wreg[kk < SQ_NRB ? kk : 0][j] : bb[kk % 3][j]),
                                                                 __builtin_bit_cast(bf16x8_t, af), acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < SQ_J; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) dhr[j][r] = acc[j][r];
        __syncthreads();                                  // all reads of the dgates tile done before the next step overwrites it
    }
#pragma unroll
    for (int j = 0; j < SQ_J; ++j)
        *reinterpret_cast<f32x4_t*>(D.dh0 + (long)row * a.lddh0 + 64 * w + 16 * j + 4 * lq) = (f32x4_t){dhr[j][0], dhr[j][1], dhr[j][2], dhr[j][3]};
}

extern "C" int cst_lstm_seq_bwd(const void* wt0, const void* wt1, const float* gates0, const float* gates1,
                                const float* cenc0, const float* cenc1, const float* c_last, long ldcl,
                                const float* dc_last, long lddcl, const float* dmem,
                                float* dgates0, float* dgates1, void* dgates0_bf16, void* dgates1_bf16,
                                float* dh0, long lddh0, int B, int L, int H, void* stream) {
    CST_REQUIRE(wt0 && wt1 && gates0 && gates1 && cenc0 && cenc1 && c_last && dc_last && dmem && dgates0 && dgates1 && dh0,
                "cst_lstm_seq_bwd: null pointer");
    CST_REQUIRE(H == SQ_H && B > 0 && B % 16 == 0 && L > 0, "cst_lstm_seq_bwd: needs H == %d and B %% 16 == 0 (H=%d, B=%d)", SQ_H, H, B);
    CST_REQUIRE(((((uintptr_t)wt0) | ((uintptr_t)wt1)) & 15) == 0, "cst_lstm_seq_bwd: W_hh^T fragment copies must be 16-byte aligned");
    LstmSeqBwdArgs a;
    CST_REQUIRE(!dgates0_bf16 == !dgates1_bf16, "cst_lstm_seq_bwd: pass both bf16 dgates twins or neither");
    {   // 16-byte loads / stores of four consecutive units per lane
        const uintptr_t al = (uintptr_t)gates0 | (uintptr_t)gates1 | (uintptr_t)cenc0 | (uintptr_t)cenc1 | (uintptr_t)c_last | (uintptr_t)dc_last |
                             (uintptr_t)dmem | (uintptr_t)dgates0 | (uintptr_t)dgates1 | (uintptr_t)dh0;
        CST_REQUIRE((al & 15) == 0 && ldcl % 4 == 0 && lddcl % 4 == 0 && lddh0 % 4 == 0 &&
                    ((((uintptr_t)dgates0_bf16) | ((uintptr_t)dgates1_bf16)) & 7) == 0,
                    "cst_lstm_seq_bwd: fp32 operands must be 16-byte aligned with leading dimensions that are multiples of 4 (bf16 twins: 8-byte)");
    }
    a.dir[0] = LstmSeqBwdDir{(const bf16_t*)wt0, gates0, cenc0, c_last, dc_last, dgates0, (bf16_t*)dgates0_bf16, dh0, 0};
    a.dir[1] = LstmSeqBwdDir{(const bf16_t*)wt1, gates1, cenc1, c_last + H, dc_last + H, dgates1, (bf16_t*)dgates1_bf16, dh0 + H, 1};
    a.dmem = dmem; a.B = B; a.L = L; a.ldcl = ldcl; a.lddcl = lddcl; a.lddh0 = lddh0;
    static CstPerDevice attr_done;
    if (cst_first_on_device(attr_done)) {
        (void)hipFuncSetAttribute((const void*)lstm_seq_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SQ_BWD_LDS);
    }
    hipLaunchKernelGGL(lstm_seq_bwd_kernel, dim3(B / 16, 2), dim3(256), SQ_BWD_LDS, (hipStream_t)stream, a);
    CST_LAUNCH_CHECK("cst_lstm_seq_bwd");
    return CST_OK;
}
