// RelGAN discriminator convolution bank, fused for gfx950 (reference: src/model/discriminator.py:21-24
// the Conv2d(1, F, (k, es), stride=(1, es)) bank, :41-44 relu + max over time).
//
// The embedded sentence e [B, L, E] is read as R = E / es "representations" of width es; sample
// g = b * R + rep sees the flat signal X_g[l * es + c] = e[b, l, rep * es + c], and filter size k is
// a 1-D correlation over X_g with window KE = k * es and stride es:
//     y[g, t, f] = relu(bias[f] + sum_kk X_g[t * es + kk] * w[f, kk]),  t < T = L - k + 1
//     feats[g, f] = max_t y[g, t, f]        arg[g, f] = first t attaining it
// The unfused formulation (im2col -> GEMM -> max) writes and re-reads the [G*T, F] plane: 84 MB per
// filter size at the bench shape (G = 4096, T = 17, F = 300), three times per step, and the backward
// materialises the same plane again with one non-zero in T.  Here the plane lives in LDS only:
//   forward      one workgroup per block of GB samples (<= 144 (g,t) rows = 9 MFMA row tiles); the A
//                fragments of all rows stay in registers, each wave walks its own 16-filter tiles on
//                the exact-fp32 matrix pipe and reduces over t through a private LDS tile.
//   input grad   dX_g = col2im(dy W) with dy rebuilt on the fly from (arg, dfeats, relu gate) as the
//                A operand; dy W per block in LDS, then the k-window overlap-add inside each g.
//   weight grad  a gather: dW[f, :] = sum_g d[g, f] * X_g[arg[g, f] * es : + KE]; per-workgroup partials
//                in a slab, summed in workgroup order by a second kernel (deterministic).
// Limits (the host wrapper falls back to the im2col path outside them): KE <= 40, es % 4 == 0,
// T <= 144, F <= 320.
#include "cst_common.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int RC_ROWS = 144;           // (g, t) rows per workgroup
constexpr int RC_MT = RC_ROWS / 16;    // MFMA row tiles
constexpr int RC_KS = 10;              // k-steps of 4 over the window: KE <= 40
constexpr int RC_GB = 16;              // samples per workgroup (forward / input grad)
constexpr int RC_FS = 80;              // k-steps of 4 over the filters: F <= 320
constexpr int RC_GC = 8;               // samples staged at a time by the weight-grad gather

struct RelConvGeom {
    int B, L, E, R, k, F;
    int es, T, KE, G;
    int GB, MT, XS;                    // samples / row tiles per workgroup, LDS stride of one X_g
};

static int relconv_geom(RelConvGeom& q, int B, int L, int E, int R, int k, int F, const char* who) {
    CST_REQUIRE(B > 0 && L > 0 && E > 0 && R > 0 && k > 0 && F > 0 && E % R == 0, "%s: bad shape", who);
    q.B = B; q.L = L; q.E = E; q.R = R; q.k = k; q.F = F;
    q.es = E / R; q.T = L - k + 1; q.KE = k * q.es; q.G = B * R;
    CST_REQUIRE(q.T >= 1, "%s: needs L >= k (L=%d, k=%d)", who, L, k);
    CST_REQUIRE(q.KE <= 4 * RC_KS && q.es % 4 == 0 && q.T <= RC_ROWS && F <= 4 * RC_FS,
                "%s: outside the fused kernel's limits (KE=%d <= 40, es=%d %% 4 == 0, T=%d <= 144, F=%d <= 320)", who, q.KE, q.es, q.T, F);
    int gb = RC_ROWS / q.T;
    if (gb > RC_GB) gb = RC_GB;
    const int want = (q.G + 511) / 512;                  // >= 2 workgroups per CU when G allows
    if (gb > want) gb = want < 1 ? 1 : want;
    q.GB = gb;
    q.MT = (gb * q.T + 15) / 16;
    q.XS = L * q.es + 8;
    return CST_OK;
}

// stage X_g for GB (or RC_GC) consecutive samples starting at g0 (samples past G replicate G-1)
__device__ __forceinline__ void relconv_stage_x(const float* __restrict__ e, float* X, const RelConvGeom& q, int g0, int ngl) {
    for (int i = threadIdx.x; i < ngl * q.XS; i += blockDim.x) {
        const int gl = i / q.XS, o = i - gl * q.XS;
        float v = 0.f;
        if (o < q.L * q.es) {
            const int g = min(g0 + gl, q.G - 1);
            const int b = g / q.R, rep = g - b * q.R;
            const int l = o / q.es, c = o - l * q.es;
            v = e[((long)b * q.L + l) * q.E + rep * q.es + c];
        }
        X[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relconv_fwd_kernel(const float* __restrict__ e, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ feats, long ldf,
                                                          int* __restrict__ arg, RelConvGeom q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                                   // [GB][XS]
    float* Cs = X + q.GB * q.XS;                       // [4 waves][RC_ROWS][17]
    const int g0 = blockIdx.x * q.GB;
    const int ng = min(q.GB, q.G - g0);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    relconv_stage_x(e, X, q, g0, q.GB);
    __syncthreads();
    // A fragments of every row tile: A[m = (gl, t)][k = 4 kk + lq] = X_gl[t * es + k]
    const int rows = q.GB * q.T;
    const int ks = (q.KE + 3) >> 2;
    float a[RC_MT][RC_KS];
#pragma unroll
    for (int mt = 0; mt < RC_MT; ++mt) {
        const int m = min(mt * 16 + lr, rows - 1);
        const int gl = m / q.T, t = m - gl * q.T;
        const float* xr = X + gl * q.XS + t * q.es + lq;   // k >= KE reads the next window positions / zero slack: finite, times a zero B
#pragma unroll
        for (int kk = 0; kk < RC_KS; ++kk) a[mt][kk] = (mt < q.MT && kk < ks) ? xr[4 * kk] : 0.f;
    }
    float* Cw = Cs + wv * RC_ROWS * 17;
    const int ntiles = (q.F + 15) >> 4;
    for (int nt = wv; nt < ntiles; nt += 4) {
        const int n = nt * 16 + lr;
        float bfr[RC_KS];
#pragma unroll
        for (int kk = 0; kk < RC_KS; ++kk) {
            const int kidx = 4 * kk + lq;
            bfr[kk] = (kk < ks && n < q.F && kidx < q.KE) ? w[(long)n * q.KE + kidx] : 0.f;
        }
        const float bv = n < q.F ? bias[n] : 0.f;
#pragma unroll
        for (int mt = 0; mt < RC_MT; ++mt) {
            if (mt < q.MT) {
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kk = 0; kk < RC_KS; ++kk)
                    if (kk < ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][kk], bfr[kk], acc, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) Cw[(mt * 16 + lq * 4 + r) * 17 + lr] = fmaxf(acc[r] + bv, 0.f);
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int gl = lq; gl < ng; gl += 4) {          // 16-lane group lq scans samples lq, lq + 4, ...; lane lr = filter
            float best = -INFINITY;
            int bi = 0;
            for (int t = 0; t < q.T; ++t) {
                const float v = Cw[(gl * q.T + t) * 17 + lr];
                if (v > best) { best = v; bi = t; }
            }
            if (n < q.F) {
                feats[(long)(g0 + gl) * ldf + n] = best;
                arg[(long)(g0 + gl) * q.F + n] = bi;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int cst_relconv_fwd(const float* e, int B, int L, int E, int R, int k, const float* w, const float* bias, int F,
                               float* feats, long ldf, int* arg, void* stream) {
    CST_REQUIRE(e && w && bias && feats && arg, "cst_relconv_fwd: null pointer");
    CST_REQUIRE(ldf >= F, "cst_relconv_fwd: ldf < F");
    RelConvGeom q;
    if (int rc = relconv_geom(q, B, L, E, R, k, F, "cst_relconv_fwd")) return rc;
    const size_t lds = sizeof(float) * ((size_t)q.GB * q.XS + 4 * RC_ROWS * 17);
    hipLaunchKernelGGL(relconv_fwd_kernel, dim3((q.G + q.GB - 1) / q.GB), dim3(256), lds, (hipStream_t)stream, e, w, bias, feats, ldf, arg, q);
    CST_LAUNCH_CHECK("cst_relconv_fwd");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// de[b, l, rep*es + c] (+)= sum_{w < k, 0 <= l-w < T} dcol[g, l-w, w*es + c],  dcol = dy W,
// dy[g, t, f] = (arg[g,f] == t && feats[g,f] > 0) ? dfeats[g,f] : 0
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relconv_bwd_input_kernel(const float* __restrict__ dfeats, long ldd,
                                                                const float* __restrict__ feats, long ldf,
                                                                const int* __restrict__ arg, const float* __restrict__ w,
                                                                float* __restrict__ de, int accumulate, RelConvGeom q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int FP = (q.F + 3) & ~3;
    const int NTk = (q.KE + 15) >> 4, KEP = NTk * 16 + 1;
    float* dsel = smem;                                             // [GB][FP]  gated upstream gradient
    int* argl = reinterpret_cast<int*>(dsel + q.GB * FP);           // [GB][FP]
    float* dcs = reinterpret_cast<float*>(argl + q.GB * FP);        // [RC_ROWS][KEP]  dy W
    const int g0 = blockIdx.x * q.GB;
    const int ng = min(q.GB, q.G - g0);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    for (int i = threadIdx.x; i < q.GB * FP; i += 256) {
        const int gl = i / FP, f = i - gl * FP;
        const int g = g0 + gl;
        const bool ok = g < q.G && f < q.F;
        dsel[i] = (ok && feats[(long)g * ldf + f] > 0.f) ? dfeats[(long)g * ldd + f] : 0.f;
        argl[i] = ok ? arg[(long)g * q.F + f] : -1;
    }
    __syncthreads();
    const int rows = q.GB * q.T;
    const int ksf = FP >> 2;
    for (int nt = 0; nt < NTk; ++nt) {
        // B[k = f][n = kk]: lane holds w[f = 4 s + lq][kk = nt*16 + lr]
        const int kk = nt * 16 + lr;
        float bfr[RC_FS];
#pragma unroll
        for (int s = 0; s < RC_FS; ++s) {
            const int f = 4 * s + lq;
            bfr[s] = (s < ksf && f < q.F && kk < q.KE) ? w[(long)f * q.KE + kk] : 0.f;
        }
        for (int mt = wv; mt < q.MT; mt += 4) {
            const int m = min(mt * 16 + lr, rows - 1);
            const int gl = m / q.T, t = m - gl * q.T;
            const float* dr = dsel + gl * FP + lq;
            const int* ar = argl + gl * FP + lq;
            f32x4_t acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < RC_FS; s += 2) {
                if (s < ksf) {
                    const float av = ar[4 * s] == t ? dr[4 * s] : 0.f;
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfr[s], acc0, 0, 0, 0);
                }
                if (s + 1 < ksf) {
                    const float av = ar[4 * s + 4] == t ? dr[4 * s + 4] : 0.f;
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bfr[s + 1], acc1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dcs[(mt * 16 + lq * 4 + r) * KEP + nt * 16 + lr] = acc0[r] + acc1[r];
        }
    }
    __syncthreads();
    const int per = q.L * q.es;
    for (int i = threadIdx.x; i < ng * per; i += 256) {
        const int gl = i / per, o = i - gl * per;
        const int l = o / q.es, c = o - l * q.es;
        float s = 0.f;
        for (int x = 0; x < q.k; ++x) {
            const int t = l - x;
            if (t >= 0 && t < q.T) s += dcs[(gl * q.T + t) * KEP + x * q.es + c];
        }
        const int g = g0 + gl;
        const int b = g / q.R, rep = g - b * q.R;
        float* dst = de + ((long)b * q.L + l) * q.E + rep * q.es + c;
        *dst = accumulate ? *dst + s : s;
    }
}

extern "C" int cst_relconv_bwd_input(const float* dfeats, long ldd, const float* feats, long ldf, const int* arg,
                                     const float* w, int B, int L, int E, int R, int k, int F,
                                     float* de, int accumulate, void* stream) {
    CST_REQUIRE(dfeats && feats && arg && w && de, "cst_relconv_bwd_input: null pointer");
    CST_REQUIRE(ldd >= F && ldf >= F, "cst_relconv_bwd_input: leading dimension < F");
    RelConvGeom q;
    if (int rc = relconv_geom(q, B, L, E, R, k, F, "cst_relconv_bwd_input")) return rc;
    const int FP = (F + 3) & ~3, KEP = ((q.KE + 15) / 16) * 16 + 1;
    const size_t lds = sizeof(float) * ((size_t)2 * q.GB * FP + (size_t)RC_ROWS * KEP);
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)relconv_bwd_input_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(relconv_bwd_input_kernel, dim3((q.G + q.GB - 1) / q.GB), dim3(256), lds, (hipStream_t)stream,
                       dfeats, ldd, feats, ldf, arg, w, de, accumulate, q);
    CST_LAUNCH_CHECK("cst_relconv_bwd_input");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// dW[f, kk] = sum_g d[g, f] X_g[arg[g, f] * es + kk],  db[f] = sum_g d[g, f]
// slab [nwg][KE + 1][F]: workgroup partials, thread = filter so slab rows are written coalesced
// ---------------------------------------------------------------------------------------------
__global__ void relconv_bwd_weight_kernel(const float* __restrict__ dfeats, long ldd, const float* __restrict__ feats, long ldf,
                                          const int* __restrict__ arg, const float* __restrict__ e,
                                          float* __restrict__ slab, int gper, RelConvGeom q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;                                   // [RC_GC][XS]
    const int gbeg = blockIdx.x * gper, gend = min(q.G, gbeg + gper);
    const int f = threadIdx.x;
    const bool fok = f < q.F;
    float acc[4 * RC_KS];
#pragma unroll
    for (int i = 0; i < 4 * RC_KS; ++i) acc[i] = 0.f;
    float dbacc = 0.f;
    const int ks = q.KE >> 2;
    for (int gc = gbeg; gc < gend; gc += RC_GC) {
        const int n = min(RC_GC, gend - gc);
        // this chunk's gradients and argmax positions first: every load in flight before the barrier
        float dv[RC_GC];
        int av[RC_GC];
#pragma unroll
        for (int gl = 0; gl < RC_GC; ++gl) {
            const int g = min(gc + gl, q.G - 1);
            const float fv = fok ? feats[(long)g * ldf + f] : 0.f;
            const float dd = fok ? dfeats[(long)g * ldd + f] : 0.f;
            av[gl] = fok ? arg[(long)g * q.F + f] : 0;
            dv[gl] = (gl < n && fv > 0.f) ? dd : 0.f;
        }
        __syncthreads();                               // previous chunk's X fully consumed
        relconv_stage_x(e, X, q, gc, RC_GC);
        __syncthreads();
#pragma unroll
        for (int gl = 0; gl < RC_GC; ++gl) {
            const float d = dv[gl];
            if (d != 0.f) {
                const float4* row = reinterpret_cast<const float4*>(X + gl * q.XS + av[gl] * q.es);
#pragma unroll
                for (int k4 = 0; k4 < RC_KS; ++k4) {
                    if (k4 < ks) {
                        const float4 x = row[k4];
                        acc[4 * k4 + 0] += d * x.x; acc[4 * k4 + 1] += d * x.y;
                        acc[4 * k4 + 2] += d * x.z; acc[4 * k4 + 3] += d * x.w;
                    }
                }
                dbacc += d;
            }
        }
    }
    if (fok) {
        float* out = slab + (long)blockIdx.x * (q.KE + 1) * q.F + f;
#pragma unroll
        for (int kk = 0; kk < 4 * RC_KS; ++kk)
            if (kk < q.KE) out[(long)kk * q.F] = acc[kk];
        out[(long)q.KE * q.F] = dbacc;
    }
}

// 256 threads = 64 outputs x 4 interleaved workgroup subsets; the four partials are added in subset order
__global__ __launch_bounds__(256) void relconv_bwd_weight_reduce(const float* __restrict__ slab, int nwg, float* __restrict__ dw,
                                                                 float* __restrict__ db, int KE, int F) {
    __shared__ float part[4][64];
    const int o = blockIdx.x * 64 + (threadIdx.x & 63), p = threadIdx.x >> 6;
    const int total = (KE + 1) * F;
    float s = 0.f;
    if (o < total)
        for (int wg = p; wg < nwg; wg += 4) s += slab[(long)wg * total + o];
    part[p][threadIdx.x & 63] = s;
    __syncthreads();
    if (p == 0 && o < total) {
        const float v = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
        const int kk = o / F, f = o - kk * F;
        if (kk < KE) dw[(long)f * KE + kk] = v;
        else db[f] = v;
    }
}

extern "C" long cst_relconv_bwd_weight_workspace_floats(int B, int R, int k, int E, int F) {
    const long G = (long)B * R;
    const long nwg = G < 256 ? G : 256;
    return nwg * ((long)k * (E / R) + 1) * F;
}

extern "C" int cst_relconv_bwd_weight(const float* dfeats, long ldd, const float* feats, long ldf, const int* arg,
                                      const float* e, int B, int L, int E, int R, int k, int F,
                                      float* dw, float* db, float* workspace, long workspace_floats, void* stream) {
    CST_REQUIRE(dfeats && feats && arg && e && dw && db && workspace, "cst_relconv_bwd_weight: null pointer");
    CST_REQUIRE(ldd >= F && ldf >= F, "cst_relconv_bwd_weight: leading dimension < F");
    RelConvGeom q;
    if (int rc = relconv_geom(q, B, L, E, R, k, F, "cst_relconv_bwd_weight")) return rc;
    const int nwg = q.G < 256 ? q.G : 256;
    const int gper = (q.G + nwg - 1) / nwg;
    const int used = (q.G + gper - 1) / gper;           // workgroups that own at least one sample
    CST_REQUIRE((long)used * (q.KE + 1) * F <= workspace_floats, "cst_relconv_bwd_weight: workspace too small (%ld floats needed)",
                (long)used * (q.KE + 1) * F);
    const int threads = (F + 63) / 64 * 64;
    const size_t lds = sizeof(float) * (size_t)RC_GC * q.XS;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(relconv_bwd_weight_kernel, dim3(used), dim3(threads), lds, st, dfeats, ldd, feats, ldf, arg, e, workspace, gper, q);
    CST_LAUNCH_CHECK("cst_relconv_bwd_weight");
    const int total = (q.KE + 1) * F;
    hipLaunchKernelGGL(relconv_bwd_weight_reduce, dim3((total + 63) / 64), dim3(256), 0, st, workspace, used, dw, db, q.KE, F);
    CST_LAUNCH_CHECK("cst_relconv_bwd_weight_reduce");
    return CST_OK;
}
