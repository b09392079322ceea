// Argument block shared by the bf16-operand GEMM kernels (gemm_bf16.hip: the LDS-DMA tile kernels; gemm_pp.hip: the 8-wave ping-pong
// kernel with big tiles).  Internal to libcst_hip.so -- the C ABI is include/cst_hip.h.
#pragma once
#include "cst_common.h"

typedef unsigned short bf16_t;

struct BGemmArgs {
    const bf16_t* A; const bf16_t* B;
    const bf16_t* A2; const bf16_t* B2;   // second independent problem of the same shape (gridDim.z == 2; slab output only)
    float* C; bf16_t* Cb;           // either or both
    const float* bias; const float* addend; const bf16_t* aux;
    const float* bscale;            // per-output-column scale (fp8 B operand: B[n][k] = fp8[n][k] * bscale[n]); null = none
    long lda, ldb, ldc, ldcb, ldadd, ldaux;
    int M, N, K;                    // K multiple of 64
    int act;                        // 0 none, 1 relu, 2 leaky(0.1), 3 aux>0 ? v*gate_scale : 0, 4 aux>0 ? v : 0.1 v
    int accumulate;                 // C += v (fp32 output only)
    float alpha, gate_scale;
    CstDrop drop;
    int splits, k_per_split;        // k_per_split multiple of 64
    int slab_only;                  // write the raw partial product(s) to the slab even when splits == 1
    float* slab;
    unsigned long long* amax;       // optional [AMAX_GROUPS][M]: packed arg-max words of every output row, folded in with 64-bit atomic max (cst_gemm_bf16_argmax)
    int gn;                         // tile columns per XCD strip (0: the default, 8)
    int abl;                        // timing ablations (CST_GB_ABL, tools/gemm_bench.py abl): 1 no DMA, 2 no MFMA, 4 no fragment reads, 8 no write-out
};

// kernel-precise timing shared with gemm.hip (cst_gemm_profile_enable / _read)
bool cst_prof_on();
void cst_prof_push(hipEvent_t a, hipEvent_t b, double flops, double bytes, int which);
void cst_prof_push_shape(hipEvent_t a, hipEvent_t b, double flops, double bytes, int which, int m, int n, int k);

// gemm_pp.hip: performs the product on the ping-pong kernel when it qualifies (returns 1) or leaves it to the caller's tile kernels (0).
// `tiles(ctx)` (optional) runs the same product on the tile kernels: the first eager call of a shape times every candidate build AND the
// tile kernels and remembers the winner (the product has then been computed: returns 1 either way).
int bgemm_pp_try(const BGemmArgs& g, int force_cfg, hipStream_t st, int (*tiles)(void*) = nullptr, void* ctx = nullptr);
