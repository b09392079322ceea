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
//   forward      one workgroup per block of GB samples (<= 128 (g,t) rows = 8 MFMA row tiles); the A
//                fragments of all rows stay in registers, each wave walks its own 16-filter tiles on
//                the exact-fp32 matrix pipe and reduces over t through a private LDS tile.
//   input grad   dX_g = col2im(dy W) with dy rebuilt on the fly from (arg, dfeats, relu gate) as the
//                A operand; dy W per block in LDS, then the k-window overlap-add inside each g.
//   weight grad  a gather: dW[f, :] = sum_g d[g, f] * X_g[arg[g, f] * es : + KE]; per-workgroup partials
//                in a slab, summed in workgroup order by a second kernel (deterministic).
// Limits (the host wrapper falls back to the im2col path outside them): KE <= 40, es % 4 == 0,
// T <= 128, F <= 320.
#include "cst_common.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int RC_ROWS = 128;           // (g, t) rows per workgroup (8 row tiles: two per wave in the input grad)
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
                "%s: outside the fused kernel's limits (KE=%d <= 40, es=%d %% 4 == 0, T=%d <= 128, F=%d <= 320)", who, q.KE, q.es, q.T, F);
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
// Loads below are unconditional on clamped indices and the bounds decide only what is stored / selected:
// a load under a per-lane condition becomes its own branch with its own s_waitcnt, i.e. one exposed
// memory round trip per element.
__device__ __forceinline__ void relconv_stage_x(const float* __restrict__ e, float* X, const RelConvGeom& q, int g0, int ngl) {
    const int n = ngl * q.XS, per = q.L * q.es, bd = blockDim.x;
    const int iters = (n + bd - 1) / bd;
#pragma unroll 4
    for (int it = 0; it < iters; ++it) {
        const int ir = threadIdx.x + it * bd;
        const int i = min(ir, n - 1);
        const int gl = i / q.XS, o = i - gl * q.XS;
        const int oc = min(o, per - 1);
        const int g = min(g0 + gl, q.G - 1);
        const int b = g / q.R, rep = g - b * q.R;
        const int l = oc / q.es, c = oc - l * q.es;
        const float v = e[((long)b * q.L + l) * q.E + rep * q.es + c];
        if (ir < n) X[i] = o < per ? v : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// KS = ceil(KE / 4) k-steps is a template parameter so that the MFMA chains carry no bounds branches
// ---------------------------------------------------------------------------------------------
template <int KS>
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
    const int ntiles = (q.F + 15) >> 4;
    // B fragments (filters nt*16 + lr) of all of this wave's <= 5 filter tiles are requested up front
    // and land while X is staged: fetched tile by tile, each tile's short MFMA burst (8*KS MFMAs)
    // would wait a full L2 round trip.
    constexpr int NTW = RC_FS / 4 / 4;                  // F <= 320 -> <= 20 tiles -> <= 5 per wave
    float ball[NTW][KS], bvall[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = (wv + 4 * j) * 16 + lr, nc = min(n, q.F - 1);
        const float* wr = w + (long)nc * q.KE;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int kidx = 4 * kk + lq;
            const float v = wr[min(kidx, q.KE - 1)];
            ball[j][kk] = (n < q.F && kidx < q.KE) ? v : 0.f;
        }
        const float bb = bias[nc];
        bvall[j] = n < q.F ? bb : 0.f;
    }
    relconv_stage_x(e, X, q, g0, q.GB);
    __syncthreads();
    // A fragments of every row tile: A[m = (gl, t)][k = 4 kk + lq] = X_gl[t * es + k]  (k >= KE: finite
    // window spill-over times a zero B)
    const int rows = q.GB * q.T;
    float a[RC_MT][KS];
#pragma unroll
    for (int mt = 0; mt < RC_MT; ++mt) {
        const int m = min(mt * 16 + lr, rows - 1);
        const int gl = m / q.T, t = m - gl * q.T;
        const float* xr = X + gl * q.XS + t * q.es + lq;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) a[mt][kk] = xr[4 * kk];
    }
    float* Cw = Cs + wv * RC_ROWS * 17;
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int nt = wv + 4 * j;
        if (nt >= ntiles) break;
        const int n = nt * 16 + lr;
        const float bv = bvall[j];
#pragma unroll
        for (int mt = 0; mt < RC_MT; ++mt) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};                // row tiles >= MT recompute the last valid rows: never read
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt][kk], ball[j][kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Cw[(mt * 16 + lq * 4 + r) * 17 + lr] = fmaxf(acc[r] + bv, 0.f);
        }
        __builtin_amdgcn_wave_barrier();
        for (int gl = lq; gl < ng; gl += 4) {          // 16-lane group lq scans samples lq, lq + 4, ...; lane lr = filter
            float best = -INFINITY;
            int bi = 0;
#pragma unroll 6
            for (int t = 0; t < q.T; ++t) {            // independent LDS reads, a cheap compare chain
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
    const dim3 grid((q.G + q.GB - 1) / q.GB), block(256);
    hipStream_t st = (hipStream_t)stream;
#define RC_FWD(KSV) case KSV: hipLaunchKernelGGL(relconv_fwd_kernel<KSV>, grid, block, lds, st, e, w, bias, feats, ldf, arg, q); break;
    switch ((q.KE + 3) / 4) {
        RC_FWD(1) RC_FWD(2) RC_FWD(3) RC_FWD(4) RC_FWD(5) RC_FWD(6) RC_FWD(7) RC_FWD(8) RC_FWD(9) RC_FWD(10)
        default: cst_set_error("cst_relconv_fwd: window %d > 40", q.KE); return CST_ERR_ARG;
    }
#undef RC_FWD
    CST_LAUNCH_CHECK("cst_relconv_fwd");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// de[b, l, rep*es + c] (+)= sum_{x < k, 0 <= l-x < T} dcol[g, l-x, x*es + c],  dcol = dy W,
// dy[g, t, f] = (arg[g,f] == t && feats[g,f] > 0) ? dfeats[g,f] : 0
// 8 waves, one 16-row tile each; the (arg, gated gradient) pairs of the block's samples and the whole
// filter matrix sit in LDS, so the K loop over the filters touches no global memory: per k-step one
// b64 read rebuilds the A value and NT b32 reads feed NT accumulator chains.
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(512) void relconv_bwd_input_kernel(const float* __restrict__ dfeats, long ldd,
                                                                const float* __restrict__ feats, long ldf,
                                                                const int* __restrict__ arg, const float* __restrict__ w,
                                                                float* __restrict__ de, int accumulate, RelConvGeom q) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KEP = NT * 16 + 1, KW = NT * 16;
    const int FP = (q.F + 3) & ~3;
    int2* pk = reinterpret_cast<int2*>(smem);                       // [GB][FP]  {arg, gated gradient bits}
    float* Ws = smem + 2 * q.GB * FP;                               // [FP][KEP]  w[f][kk], zero past KE / F
    float* dcs = Ws;                                                // [RC_ROWS][KEP]  dy W (after the K loop)
    const int g0 = blockIdx.x * q.GB;
    const int ng = min(q.GB, q.G - g0);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    {
        const int n = q.GB * FP, iters = (n + 511) / 512;
#pragma unroll 2
        for (int it = 0; it < iters; ++it) {                 // unconditional (clamped) loads
            const int ir = threadIdx.x + it * 512;
            const int i = min(ir, n - 1);
            const int gl = i / FP, f = i - gl * FP;
            const int g = min(g0 + gl, q.G - 1), fc = min(f, q.F - 1);
            const float fv = feats[(long)g * ldf + fc];
            const float dv = dfeats[(long)g * ldd + fc];
            const int av = arg[(long)g * q.F + fc];
            const bool ok = g0 + gl < q.G && f < q.F;
            if (ir < n) pk[i] = make_int2(ok ? av : -1, __float_as_int((ok && fv > 0.f) ? dv : 0.f));
        }
        const int nw = FP * KW, itw = (nw + 511) / 512;
#pragma unroll 4
        for (int it = 0; it < itw; ++it) {
            const int ir = threadIdx.x + it * 512;
            const int i = min(ir, nw - 1);
            const int f = i / KW, kk = i - f * KW;
            const float v = w[(long)min(f, q.F - 1) * q.KE + min(kk, q.KE - 1)];
            if (ir < nw) Ws[f * KEP + kk] = (f < q.F && kk < q.KE) ? v : 0.f;
        }
    }
    __syncthreads();
    const int rows = q.GB * q.T;
    const int m = min(wv * 16 + lr, rows - 1);          // row tiles >= MT recompute the last valid rows into unused dcs rows
    const int gl = m / q.T, t = m - gl * q.T;
    const int2* pr = pk + gl * FP + lq;
    const float* br = Ws + lq * KEP + lr;
    f32x4_t acc[NT][2];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[j][c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const int ksf = FP >> 2;
    auto kstep = [&](int s2, int chain) {
        const int2 p = pr[4 * s2];
        const float av = p.x == t ? __int_as_float(p.y) : 0.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float bv = br[4 * s2 * KEP + j * 16];
            if (chain == 0) acc[j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j][0], 0, 0, 0);
            else acc[j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[j][1], 0, 0, 0);
        }
    };
    int s2 = 0;
    for (; s2 + 4 <= ksf; s2 += 4) {                    // 4 k-steps per trip: 4 b64 + 4 NT b32 reads in flight
        kstep(s2, 0); kstep(s2 + 1, 1); kstep(s2 + 2, 0); kstep(s2 + 3, 1);
    }
    for (; s2 < ksf; ++s2) kstep(s2, 0);
    __syncthreads();                                    // every wave is done with Ws before dcs overwrites it
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) dcs[(wv * 16 + lq * 4 + r) * KEP + j * 16 + lr] = acc[j][0][r] + acc[j][1][r];
    __syncthreads();
    const int per = q.L * q.es, nout = ng * per, ito = (nout + 511) / 512;
#pragma unroll 2
    for (int it = 0; it < ito; ++it) {
        const int ir = threadIdx.x + it * 512;
        const int i = min(ir, nout - 1);
        const int gi = i / per, o = i - gi * per;
        const int l = o / q.es, c = o - l * q.es;
        const int g = g0 + gi;
        const int b = g / q.R, rep = g - b * q.R;
        float* dst = de + ((long)b * q.L + l) * q.E + rep * q.es + c;
        const float prev = accumulate ? *dst : 0.f;
        float sacc = 0.f;
        for (int x = 0; x < q.k; ++x) {
            const int tt = l - x;
            if (tt >= 0 && tt < q.T) sacc += dcs[(gi * q.T + tt) * KEP + x * q.es + c];
        }
        if (ir < nout) *dst = prev + sacc;
    }
}

extern "C" int cst_relconv_bwd_input(const float* dfeats, long ldd, const float* feats, long ldf, const int* arg,
                                     const float* w, int B, int L, int E, int R, int k, int F,
                                     float* de, int accumulate, void* stream) {
    CST_REQUIRE(dfeats && feats && arg && w && de, "cst_relconv_bwd_input: null pointer");
    CST_REQUIRE(ldd >= F && ldf >= F, "cst_relconv_bwd_input: leading dimension < F");
    RelConvGeom q;
    if (int rc = relconv_geom(q, B, L, E, R, k, F, "cst_relconv_bwd_input")) return rc;
    const int FP = (F + 3) & ~3, NT = (q.KE + 15) / 16, KEP = NT * 16 + 1;
    const int wrows = FP > RC_ROWS ? FP : RC_ROWS;
    const size_t lds = sizeof(float) * ((size_t)2 * q.GB * FP + (size_t)wrows * KEP);
    const dim3 grid((q.G + q.GB - 1) / q.GB), block(512);
    hipStream_t st = (hipStream_t)stream;
#define RC_BWD(NTV)                                                                                                   \
    case NTV:                                                                                                         \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)relconv_bwd_input_kernel<NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(relconv_bwd_input_kernel<NTV>, grid, block, lds, st, dfeats, ldd, feats, ldf, arg, w, de, accumulate, q); \
        break;
    switch (NT) {
        RC_BWD(1) RC_BWD(2) RC_BWD(3)
        default: cst_set_error("cst_relconv_bwd_input: window %d > 40", q.KE); return CST_ERR_ARG;
    }
#undef RC_BWD
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
    const int fcl = min(f, q.F - 1);
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
            const float fv = feats[(long)g * ldf + fcl];
            const float dd = dfeats[(long)g * ldd + fcl];
            av[gl] = arg[(long)g * q.F + fcl];
            dv[gl] = (fok && gl < n && fv > 0.f) ? dd : 0.f;
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
    if (o < total) {
#pragma unroll 8
        for (int wg = p; wg < nwg; wg += 4) s += slab[(long)wg * total + o];
    }
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
