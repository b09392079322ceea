// Element-wise, gather/scatter, pooling, loss and optimiser kernels (all HBM-bound):
//   LSTM cell forward/backward (nn.LSTM at rnn.py:25-33, one step at a time, rnn.py:62,74)
//   embedding gather / scatter-add (+ dropout), token+position+segment embedding
//     (rnn.py:59,95; mlm.py:27-38; match.py:24-34; classifier.py:25; discriminator.py:39 one-hot path)
//   im2col / col2im for TextCNN (classifier.py:18,30) and RelGAN_D (discriminator.py:21-24,41)
//   max over time / sequence with argmax (classifier.py:32, discriminator.py:42, match.py:41)
//   highway gate (discriminator.py:45-46), dropout, activation gates
//   MSE / BCE-with-logits losses (main_pretrain.py:72; main_optimize.py:107-108,122-123)
//   global-norm clipping and Adam (Trainer gradient_clip_val + torch.optim.Adam,
//     main_pretrain.py:61-64,139; main_warmup.py:41-43,103; main_optimize.py:73-88,211)
#include "cst_common.h"

#define EW_THREADS 256
typedef unsigned short bf16_t;
__device__ __forceinline__ bf16_t pw_f2bf(float f) { __bf16 b = (__bf16)f; return __builtin_bit_cast(unsigned short, b); }
static inline dim3 ew_grid(long n) {
    long b = (n + EW_THREADS - 1) / EW_THREADS;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return dim3((unsigned)b);
}
#define EW_LOOP(i, n) for (long i = (long)blockIdx.x * EW_THREADS + threadIdx.x; i < (n); i += (long)gridDim.x * EW_THREADS)

// ---------------------------------------------------------------------------------------------
// LSTM cell.  gates [B,4H] pre-activation in, activated (i,f,g,o) out (in place, kept for bwd).
// ---------------------------------------------------------------------------------------------
__global__ void lstm_cell_fwd_kernel(float* __restrict__ gates, long ldg, const float* __restrict__ c_prev, long ldcp,
                                     float* __restrict__ h_out, long ldh, float* __restrict__ c_out, long ldc,
                                     float* __restrict__ h_out2, long ldh2,
                                     bf16_t* __restrict__ hb, long ldhb, bf16_t* __restrict__ hb2, long ldhb2, int B, int H) {
    EW_LOOP(e, (long)B * H) {
        const long b = e / H, j = e % H;
        float* g = gates + b * ldg;
        const float i = 1.f / (1.f + expf(-g[j]));
        const float f = 1.f / (1.f + expf(-g[H + j]));
        const float gg = tanhf(g[2 * H + j]);
        const float o = 1.f / (1.f + expf(-g[3 * H + j]));
        const float c = f * c_prev[b * ldcp + j] + i * gg;
        const float h = o * tanhf(c);
        g[j] = i; g[H + j] = f; g[2 * H + j] = gg; g[3 * H + j] = o;
        c_out[b * ldc + j] = c;
        h_out[b * ldh + j] = h;
        if (h_out2) h_out2[b * ldh2 + j] = h;
        if (hb) hb[b * ldhb + j] = pw_f2bf(h);             // bf16 copies: A operands of the next step's GEMMs
        if (hb2) hb2[b * ldhb2 + j] = pw_f2bf(h);
    }
}

extern "C" int cst_lstm_cell_fwd(float* gates, long ldg, const float* c_prev, long ldcp, float* h_out, long ldh,
                                 float* c_out, long ldc, float* h_out2, long ldh2,
                                 void* h_bf16, long ldhb, void* h_bf16_2, long ldhb2, int B, int H, void* stream) {
    CST_REQUIRE(gates && c_prev && h_out && c_out && B > 0 && H > 0, "cst_lstm_cell_fwd: bad arguments");
    hipLaunchKernelGGL(lstm_cell_fwd_kernel, ew_grid((long)B * H), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       gates, ldg, c_prev, ldcp, h_out, ldh, c_out, ldc, h_out2, ldh2, (bf16_t*)h_bf16, ldhb, (bf16_t*)h_bf16_2, ldhb2, B, H);
    CST_LAUNCH_CHECK("cst_lstm_cell_fwd");
    return CST_OK;
}

// dh: total gradient wrt h_t (up to two addends); dc: gradient wrt c_t flowing back from step t+1
// (null = 0).  Writes dgates (pre-activation) and dc_prev.
__global__ void lstm_cell_bwd_kernel(const float* __restrict__ gates, long ldg, const float* __restrict__ c_prev, long ldcp,
                                     const float* __restrict__ c_new, long ldcn,
                                     const float* __restrict__ dh, long lddh, const float* __restrict__ dh2, long lddh2,
                                     const float* dc, long lddc, float* __restrict__ dgates, long lddg,
                                     float* dc_prev, long lddcp, bf16_t* __restrict__ dgb, long lddgb, int B, int H) {
    EW_LOOP(e, (long)B * H) {
        const long b = e / H, j = e % H;
        const float* g = gates + b * ldg;
        const float i = g[j], f = g[H + j], gg = g[2 * H + j], o = g[3 * H + j];
        const float tc = tanhf(c_new[b * ldcn + j]);
        float dht = dh ? dh[b * lddh + j] : 0.f;
        if (dh2) dht += dh2[b * lddh2 + j];
        const float dct = (dc ? dc[b * lddc + j] : 0.f) + dht * o * (1.f - tc * tc);
        float* dg = dgates + b * lddg;
        const float g0 = dct * gg * i * (1.f - i), g1 = dct * c_prev[b * ldcp + j] * f * (1.f - f);
        const float g2 = dct * i * (1.f - gg * gg), g3 = dht * tc * o * (1.f - o);
        dg[j] = g0; dg[H + j] = g1; dg[2 * H + j] = g2; dg[3 * H + j] = g3;
        if (dgb) {
            bf16_t* q = dgb + b * lddgb;
            q[j] = pw_f2bf(g0); q[H + j] = pw_f2bf(g1); q[2 * H + j] = pw_f2bf(g2); q[3 * H + j] = pw_f2bf(g3);
        }
        dc_prev[b * lddcp + j] = dct * f;
    }
}

extern "C" int cst_lstm_cell_bwd(const float* gates, long ldg, const float* c_prev, long ldcp, const float* c_new, long ldcn,
                                 const float* dh, long lddh, const float* dh2, long lddh2, const float* dc, long lddc,
                                 float* dgates, long lddg, float* dc_prev, long lddcp, void* dgates_bf16, long lddgb,
                                 int B, int H, void* stream) {
    CST_REQUIRE(gates && c_prev && c_new && dgates && dc_prev && B > 0 && H > 0, "cst_lstm_cell_bwd: bad arguments");
    hipLaunchKernelGGL(lstm_cell_bwd_kernel, ew_grid((long)B * H), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       gates, ldg, c_prev, ldcp, c_new, ldcn, dh, lddh, dh2, lddh2, dc, lddc, dgates, lddg, dc_prev, lddcp, (bf16_t*)dgates_bf16, lddgb, B, H);
    CST_LAUNCH_CHECK("cst_lstm_cell_bwd");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// embedding gather: out[r, :] = table[select(r)] (* dropout).  select(r) = coin ? ids_a[r] : ids_b[r*ldb]
// (scheduled-sampling choice of rnn.py:91-95 made on device so a captured graph stays static).
// transposed != 0 reads table[c * ldt + id] (columns of RelGAN_D's Linear weight, one-hot path).
// ---------------------------------------------------------------------------------------------
__global__ void embed_gather_kernel(const int64_t* __restrict__ ids_a, const int64_t* __restrict__ ids_b, long ldb,
                                    const int* __restrict__ coin, const float* __restrict__ table, long ldt, int transposed,
                                    float* __restrict__ out, long ldo, bf16_t* __restrict__ outb, long ldob,
                                    int R, int E, int V, CstDrop drop) {
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    EW_LOOP(e, (long)R * E) {
        const long r = e / E, c = e % E;
        long id = ids_a ? ids_a[r] : 0;
        if (ids_b && !(coin && *coin)) id = ids_b[r * ldb];
        float v = 0.f;
        if (id >= 0 && id < V) v = transposed ? table[c * ldt + id] : table[id * ldt + c];
        if (drop.p > 0.f) v *= cst_drop_mask(drop, dseed, (uint32_t)e);
        out[r * ldo + c] = v;
        if (outb) outb[r * ldob + c] = pw_f2bf(v);
    }
}

extern "C" int cst_embed_gather(const int64_t* ids_a, const int64_t* ids_b, long ldb, const int* coin_dev,
                                const float* table, long ldt, int transposed, float* out, long ldo,
                                void* out_bf16, long ldob, int R, int E, int V,
                                float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream) {
    CST_REQUIRE((ids_a || ids_b) && table && out && R > 0 && E > 0, "cst_embed_gather: bad arguments");
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)R * E);
    hipLaunchKernelGGL(embed_gather_kernel, ew_grid((long)R * E), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       ids_a, ids_b, ldb, coin_dev, table, ldt, transposed, out, ldo, (bf16_t*)out_bf16, ldob, R, E, V, dr);
    CST_LAUNCH_CHECK("cst_embed_gather");
    return CST_OK;
}

__global__ void embed_scatter_add_kernel(const int64_t* __restrict__ ids_a, const int64_t* __restrict__ ids_b, long ldb,
                                         const int* __restrict__ coin, const float* __restrict__ dout, long ldo,
                                         float* __restrict__ dtable, long ldt, int transposed, int R, int E, int V, CstDrop drop) {
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    EW_LOOP(e, (long)R * E) {
        const long r = e / E, c = e % E;
        long id = ids_a ? ids_a[r] : 0;
        if (ids_b && !(coin && *coin)) id = ids_b[r * ldb];
        if (id < 0 || id >= V) continue;
        float g = dout[r * ldo + c];
        if (drop.p > 0.f) g *= cst_drop_mask(drop, dseed, (uint32_t)e);
        atomicAdd(transposed ? &dtable[c * ldt + id] : &dtable[id * ldt + c], g);
    }
}

extern "C" int cst_embed_scatter_add(const int64_t* ids_a, const int64_t* ids_b, long ldb, const int* coin_dev,
                                     const float* dout, long ldo, float* dtable, long ldt, int transposed, int R, int E, int V,
                                     float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream) {
    CST_REQUIRE((ids_a || ids_b) && dout && dtable && R > 0 && E > 0, "cst_embed_scatter_add: bad arguments");
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)R * E);
    hipLaunchKernelGGL(embed_scatter_add_kernel, ew_grid((long)R * E), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       ids_a, ids_b, ldb, coin_dev, dout, ldo, dtable, ldt, transposed, R, E, V, dr);
    CST_LAUNCH_CHECK("cst_embed_scatter_add");
    return CST_OK;
}

// The decoder's embedding gradients of ALL steps in one launch (rnn.py:88-96 backward): row r = s*B + b of
// dout is the gradient of the token fed to step s+1, chosen by coins[s] between the fed-back id ids_a[s*B+b]
// and the teacher token ids_b[b*ldb + s]; its dropout mask is that of call-site stream drop.stream + s, element
// b*E + c -- exactly what S separate cst_embed_scatter_add launches (one per step) compute.
__global__ void embed_scatter_steps_kernel(const int64_t* __restrict__ ids_a, const int64_t* __restrict__ ids_b, long ldb,
                                           const int* __restrict__ coins, const float* __restrict__ dout, long ldo,
                                           float* __restrict__ dtable, long ldt, int S, int B, int E, int V, CstDrop drop) {
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    const int total = S * B * E;                       // < 2^31 (checked by the host wrapper): 32-bit index arithmetic
    for (int e = blockIdx.x * EW_THREADS + threadIdx.x; e < total; e += gridDim.x * EW_THREADS) {
        const int r = e / E, c = e - r * E;
        const int s = r / B, b = r - s * B;
        long id = ids_a[r];
        if (ids_b && !(coins && coins[s])) id = ids_b[(long)b * ldb + s];
        if (id < 0 || id >= V) continue;
        float g = dout[(long)r * ldo + c];
        if (drop.p > 0.f)
            g *= ((cst_mix32(dseed, drop.stream + (uint32_t)s, (uint32_t)(b * E + c) + drop.base) >> 8) >= drop.thresh) ? drop.scale : 0.0f;
        atomicAdd(&dtable[id * ldt + c], g);
    }
}

extern "C" int cst_embed_scatter_add_steps(const int64_t* ids_a, const int64_t* ids_b, long ldb, const int* coins_dev,
                                           const float* dout, long ldo, float* dtable, long ldt, int S, int B, int E, int V,
                                           float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream) {
    CST_REQUIRE(ids_a && dout && dtable && S > 0 && B > 0 && E > 0 && ldo >= E && ldt >= E, "cst_embed_scatter_add_steps: bad arguments");
    CST_REQUIRE((long)S * B * E < (1L << 31), "cst_embed_scatter_add_steps: S*B*E too large");
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)B * E);
    hipLaunchKernelGGL(embed_scatter_steps_kernel, ew_grid((long)S * B * E), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       ids_a, ids_b, ldb, coins_dev, dout, ldo, dtable, ldt, S, B, E, V, dr);
    CST_LAUNCH_CHECK("cst_embed_scatter_add_steps");
    return CST_OK;
}

// token (+pre-multiplied soft) + position + segment embedding into rows [off, off+L) of x [B,S,d]
__global__ void tps_embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ pre,
                                     const float* __restrict__ Etok, const float* __restrict__ Epos, const float* __restrict__ seg,
                                     float* __restrict__ x, int B, int L, int d, int S, int off, int V) {
    EW_LOOP(e, (long)B * L * d) {
        const long c = e % d, l = (e / d) % L, b = e / ((long)d * L);
        float v = Epos[l * d + c];
        if (seg) v += seg[c];
        if (ids) { const long id = ids[b * L + l]; if (id >= 0 && id < V) v += Etok[id * d + c]; }
        else v += pre[(b * L + l) * d + c];
        x[(b * S + off + l) * d + c] = v;
    }
}

extern "C" int cst_tps_embed_fwd(const int64_t* ids, const float* pre, const float* Etok, const float* Epos, const float* seg_row,
                                 float* x, int B, int L, int d, int S, int off, int V, void* stream) {
    CST_REQUIRE((ids || pre) && Epos && x && B > 0 && L > 0 && d > 0 && off + L <= S, "cst_tps_embed_fwd: bad arguments");
    CST_REQUIRE(!ids || Etok, "cst_tps_embed_fwd: ids need the token table");
    hipLaunchKernelGGL(tps_embed_fwd_kernel, ew_grid((long)B * L * d), dim3(EW_THREADS), 0, (hipStream_t)stream,
                       ids, pre, Etok, Epos, seg_row, x, B, L, d, S, off, V);
    CST_LAUNCH_CHECK("cst_tps_embed_fwd");
    return CST_OK;
}

// backward: dEpos[l] += sum_b dx[b,off+l]; dseg += sum_{b,l}; dEtok scatter-add (ids) or dpre copy
__global__ __launch_bounds__(256) void tps_embed_bwd_kernel(const float* __restrict__ dx, const int64_t* __restrict__ ids,
                                                            float* __restrict__ dpre, float* __restrict__ dEtok,
                                                            float* __restrict__ dEpos, float* __restrict__ dseg,
                                                            int B, int L, int d, int S, int off, int V) {
    __shared__ float sh[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int l = blockIdx.y;
    // blockIdx.z owns a slice of the batch rows: enough workgroups to fill the chip; the position / segment sums of
    // the slices meet in atomics (the token table already does)
    const int per = (B + gridDim.z - 1) / gridDim.z;
    const int b0 = blockIdx.z * per, b1 = min(B, b0 + per);
    float s = 0.f;
    if (c < d) {
        for (int b = b0 + w; b < b1; b += 4) {
            const float g = dx[((long)b * S + off + l) * d + c];
            s += g;
            if (ids) { const long id = ids[(long)b * L + l]; if (dEtok && id >= 0 && id < V) atomicAdd(&dEtok[id * d + c], g); }
            else if (dpre) dpre[((long)b * L + l) * d + c] = g;
        }
    }
    sh[w][lane] = s;
    __syncthreads();
    if (w == 0 && c < d) {
        const float t = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
        if (dEpos) atomicAdd(&dEpos[(long)l * d + c], t);
        if (dseg) atomicAdd(&dseg[c], t);
    }
}

extern "C" int cst_tps_embed_bwd(const float* dx, const int64_t* ids, float* dpre, float* dEtok, float* dEpos, float* dseg_row,
                                 int B, int L, int d, int S, int off, int V, void* stream) {
    CST_REQUIRE(dx && B > 0 && L > 0 && d > 0 && off + L <= S, "cst_tps_embed_bwd: bad arguments");
    const int zs = B >= 64 ? cst_div_up(B, 32) : 1;
    hipLaunchKernelGGL(tps_embed_bwd_kernel, dim3(cst_div_up(d, 64), L, zs), dim3(256), 0, (hipStream_t)stream,
                       dx, ids, dpre, dEtok, dEpos, dseg_row, B, L, d, S, off, V);
    CST_LAUNCH_CHECK("cst_tps_embed_bwd");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// im2col.  mode 0 (TextCNN): e [B,L,E], window k rows, zero padding k-1 both ends,
//   col[(b*T + t), r*E + c] = e[b, t + r - (k-1), c], T = L + k - 1.
// mode 1 (RelGAN_D): e [B,L,R*es], col[((b*R + rep)*T + t), i*es + q] = e[b, t+i, rep*es + q], T = L-k+1.
// ---------------------------------------------------------------------------------------------
template <typename OUT> __device__ __forceinline__ OUT pw_out(float v);
template <> __device__ __forceinline__ float pw_out<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t pw_out<bf16_t>(float v) { return pw_f2bf(v); }

template <typename OUT>
__global__ void im2col_kernel(const float* __restrict__ e, OUT* __restrict__ col, int B, int L, int E, int k, int mode, int R) {
    if (mode == 0) {
        const int T = L + k - 1;
        const long KE = (long)k * E;
        EW_LOOP(i, (long)B * T * KE) {
            const long cc = i % KE, t = (i / KE) % T, b = i / (KE * T);
            const long r = cc / E, c = cc % E;
            const long l = t + r - (k - 1);
            col[i] = pw_out<OUT>((l >= 0 && l < L) ? e[(b * L + l) * E + c] : 0.f);
        }
    } else {
        const int T = L - k + 1, es = E / R;
        const long KE = (long)k * es;
        EW_LOOP(i, (long)B * R * T * KE) {
            const long cc = i % KE, t = (i / KE) % T, rep = (i / (KE * T)) % R, b = i / (KE * T * R);
            const long w = cc / es, q = cc % es;
            col[i] = pw_out<OUT>(e[(b * L + t + w) * E + rep * es + q]);
        }
    }
}

extern "C" int cst_im2col(const float* e, float* col, int B, int L, int E, int k, int mode, int R, void* stream) {
    CST_REQUIRE(e && col && B > 0 && L > 0 && E > 0 && k > 0, "cst_im2col: bad arguments");
    CST_REQUIRE(mode == 0 || (mode == 1 && R > 0 && E % R == 0 && L >= k), "cst_im2col: mode 1 needs L >= k and E %% R == 0");
    const long n = mode == 0 ? (long)B * (L + k - 1) * k * E : (long)B * R * (L - k + 1) * k * (E / R);
    hipLaunchKernelGGL(im2col_kernel<float>, ew_grid(n), dim3(EW_THREADS), 0, (hipStream_t)stream, e, col, B, L, E, k, mode, R);
    CST_LAUNCH_CHECK("cst_im2col");
    return CST_OK;
}

/* the same rows in bf16 (dense, row length k E or k E / R): the A operand of the bf16 GEMM and the B operand of the transposed-read
 * weight-gradient product, written once instead of staged from fp32 by every product that reads them */
extern "C" int cst_im2col_b(const float* e, void* col_bf16, int B, int L, int E, int k, int mode, int R, void* stream) {
    CST_REQUIRE(e && col_bf16 && B > 0 && L > 0 && E > 0 && k > 0, "cst_im2col_b: bad arguments");
    CST_REQUIRE(mode == 0 || (mode == 1 && R > 0 && E % R == 0 && L >= k), "cst_im2col_b: mode 1 needs L >= k and E %% R == 0");
    const long n = mode == 0 ? (long)B * (L + k - 1) * k * E : (long)B * R * (L - k + 1) * k * (E / R);
    hipLaunchKernelGGL(im2col_kernel<bf16_t>, ew_grid(n), dim3(EW_THREADS), 0, (hipStream_t)stream, e, (bf16_t*)col_bf16, B, L, E, k, mode, R);
    CST_LAUNCH_CHECK("cst_im2col_b");
    return CST_OK;
}

// col2im: de (+)= gather of dcol (each input element sums the <= k windows that cover it)
__global__ void col2im_kernel(const float* __restrict__ dcol, float* __restrict__ de, int B, int L, int E, int k, int mode, int R, int accumulate) {
    EW_LOOP(i, (long)B * L * E) {
        const long c = i % E, l = (i / E) % L, b = i / ((long)E * L);
        float s = 0.f;
        if (mode == 0) {
            const int T = L + k - 1;
            for (int r = 0; r < k; ++r) {
                const long t = l - r + (k - 1);
                if (t >= 0 && t < T) s += dcol[((b * T + t) * k + r) * E + c];
            }
        } else {
            const int T = L - k + 1, es = E / R;
            const long rep = c / es, q = c % es;
            for (int w = 0; w < k; ++w) {
                const long t = l - w;
                if (t >= 0 && t < T) s += dcol[(((b * R + rep) * T + t) * k + w) * es + q];
            }
        }
        de[i] = accumulate ? de[i] + s : s;
    }
}

extern "C" int cst_col2im(const float* dcol, float* de, int B, int L, int E, int k, int mode, int R, int accumulate, void* stream) {
    CST_REQUIRE(dcol && de && B > 0 && L > 0 && E > 0 && k > 0, "cst_col2im: bad arguments");
    hipLaunchKernelGGL(col2im_kernel, ew_grid((long)B * L * E), dim3(EW_THREADS), 0, (hipStream_t)stream, dcol, de, B, L, E, k, mode, R, accumulate);
    CST_LAUNCH_CHECK("cst_col2im");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// max over the middle axis: x [G,T,F] -> out[g*ldo + f], arg[g*F + f] (first maximum)
// ---------------------------------------------------------------------------------------------
__global__ void seqmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ out, long ldo, int* __restrict__ arg, int G, int T, int F) {
    EW_LOOP(i, (long)G * F) {
        const long f = i % F, g = i / F;
        float bv = -INFINITY; int bi = 0;
        for (int t = 0; t < T; ++t) {
            const float v = x[(g * T + t) * F + f];
            if (v > bv) { bv = v; bi = t; }
        }
        out[g * ldo + f] = bv;
        arg[i] = bi;
    }
}

extern "C" int cst_seqmax_fwd(const float* x, float* out, long ldo, int* arg, int G, int T, int F, void* stream) {
    CST_REQUIRE(x && out && arg && G > 0 && T > 0 && F > 0 && ldo >= F, "cst_seqmax_fwd: bad arguments");
    hipLaunchKernelGGL(seqmax_fwd_kernel, ew_grid((long)G * F), dim3(EW_THREADS), 0, (hipStream_t)stream, x, out, ldo, arg, G, T, F);
    CST_LAUNCH_CHECK("cst_seqmax_fwd");
    return CST_OK;
}

// dx[g,t,f] = (t == arg[g,f] && (!relu_gate || y[g,f] > 0)) ? dout[g*ldd + f] : 0   (writes every element)
template <typename OUT>
__global__ void seqmax_bwd_kernel(const float* __restrict__ dout, long ldd, const int* __restrict__ arg,
                                  const float* __restrict__ y, long ldy, int relu_gate,
                                  OUT* __restrict__ dx, int G, int T, int F) {
    EW_LOOP(i, (long)G * T * F) {
        const long f = i % F, t = (i / F) % T, g = i / ((long)F * T);
        float v = 0.f;
        if (arg[g * F + f] == t && (!relu_gate || y[g * ldy + f] > 0.f)) v = dout[g * ldd + f];
        dx[i] = pw_out<OUT>(v);
    }
}

extern "C" int cst_seqmax_bwd(const float* dout, long ldd, const int* arg, const float* y, long ldy, int relu_gate,
                              float* dx, int G, int T, int F, void* stream) {
    CST_REQUIRE(dout && arg && dx && G > 0 && T > 0 && F > 0, "cst_seqmax_bwd: bad arguments");
    CST_REQUIRE(!relu_gate || y, "cst_seqmax_bwd: relu gate needs the pooled values");
    hipLaunchKernelGGL(seqmax_bwd_kernel<float>, ew_grid((long)G * T * F), dim3(EW_THREADS), 0, (hipStream_t)stream, dout, ldd, arg, y, ldy, relu_gate, dx, G, T, F);
    CST_LAUNCH_CHECK("cst_seqmax_bwd");
    return CST_OK;
}

/* the same gradient written in bf16 only ([G T, F] dense): operand of the bf16 dgrad / weight-gradient products behind it */
extern "C" int cst_seqmax_bwd_b(const float* dout, long ldd, const int* arg, const float* y, long ldy, int relu_gate,
                                void* dx_bf16, int G, int T, int F, void* stream) {
    CST_REQUIRE(dout && arg && dx_bf16 && G > 0 && T > 0 && F > 0, "cst_seqmax_bwd_b: bad arguments");
    CST_REQUIRE(!relu_gate || y, "cst_seqmax_bwd_b: relu gate needs the pooled values");
    hipLaunchKernelGGL(seqmax_bwd_kernel<bf16_t>, ew_grid((long)G * T * F), dim3(EW_THREADS), 0, (hipStream_t)stream, dout, ldd, arg, y, ldy, relu_gate,
                       (bf16_t*)dx_bf16, G, T, F);
    CST_LAUNCH_CHECK("cst_seqmax_bwd_b");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// small element-wise ops
// ---------------------------------------------------------------------------------------------
__global__ void dropout_kernel(const float* __restrict__ x, long ldx, float* __restrict__ out, long ldo, int R, int C, CstDrop drop) {
    const uint32_t dseed = drop.p > 0.f ? cst_drop_seed(drop) : 0u;
    EW_LOOP(i, (long)R * C) {
        const long r = i / C, c = i % C;
        float v = x[r * ldx + c];
        if (drop.p > 0.f) v *= cst_drop_mask(drop, dseed, (uint32_t)i);
        out[r * ldo + c] = v;
    }
}

extern "C" int cst_dropout(const float* x, long ldx, float* out, long ldo, int R, int C,
                           float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream) {
    CST_REQUIRE(x && out && R > 0 && C > 0, "cst_dropout: bad arguments");
    CstDrop dr = cst_make_drop(drop_p, drop_seed, drop_stream, drop_seed_dev, (long)R * C);
    hipLaunchKernelGGL(dropout_kernel, ew_grid((long)R * C), dim3(EW_THREADS), 0, (hipStream_t)stream, x, ldx, out, ldo, R, C, dr);
    CST_LAUNCH_CHECK("cst_dropout");
    return CST_OK;
}

// out = alpha*a + beta*b (b may be null), strided 2-D
__global__ void axpby_kernel(const float* __restrict__ a, long lda, float alpha, const float* __restrict__ b, long ldb, float beta,
                             float* __restrict__ out, long ldo, int R, int C) {
    EW_LOOP(i, (long)R * C) {
        const long r = i / C, c = i % C;
        float v = alpha * a[r * lda + c];
        if (b) v += beta * b[r * ldb + c];
        out[r * ldo + c] = v;
    }
}

extern "C" int cst_axpby(const float* a, long lda, float alpha, const float* b, long ldb, float beta,
                         float* out, long ldo, int R, int C, void* stream) {
    CST_REQUIRE(a && out && R > 0 && C > 0, "cst_axpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, ew_grid((long)R * C), dim3(EW_THREADS), 0, (hipStream_t)stream, a, lda, alpha, b, ldb, beta, out, ldo, R, C);
    CST_LAUNCH_CHECK("cst_axpby");
    return CST_OK;
}

// out = x * (*s) with s a device scalar (chain rule through a scalar loss without a host sync)
__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ out, long n) {
    const float f = *s;
    EW_LOOP(i, n) out[i] = x[i] * f;
}

extern "C" int cst_scale_dev(const float* x, const float* s_dev, float* out, long n, void* stream) {
    CST_REQUIRE(x && s_dev && out && n > 0, "cst_scale_dev: bad arguments");
    hipLaunchKernelGGL(scale_dev_kernel, ew_grid(n), dim3(EW_THREADS), 0, (hipStream_t)stream, x, s_dev, out, n);
    CST_LAUNCH_CHECK("cst_scale_dev");
    return CST_OK;
}

// dx = y > 0 ? dy * pos_scale : slope * dy   (slope 0 = relu, 0.1 = LeakyReLU(0.1) of rnn.py:41;
// pos_scale = 1/(1-p) when y is a relu output that was dropped out afterwards)
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float slope, float pos_scale,
                               float* __restrict__ dx, long n) {
    EW_LOOP(i, n) dx[i] = y[i] > 0.f ? dy[i] * pos_scale : slope * dy[i];
}

extern "C" int cst_act_bwd(const float* dy, const float* y, float slope, float pos_scale, float* dx, long n, void* stream) {
    CST_REQUIRE(dy && y && dx && n > 0, "cst_act_bwd: bad arguments");
    hipLaunchKernelGGL(act_bwd_kernel, ew_grid(n), dim3(EW_THREADS), 0, (hipStream_t)stream, dy, y, slope, pos_scale, dx, n);
    CST_LAUNCH_CHECK("cst_act_bwd");
    return CST_OK;
}

// highway (discriminator.py:45-46): out = sig(h)*relu(h) + (1-sig(h))*pred
__global__ void highway_fwd_kernel(const float* __restrict__ h, const float* __restrict__ pred, float* __restrict__ out, long n) {
    EW_LOOP(i, n) {
        const float s = 1.f / (1.f + expf(-h[i]));
        out[i] = s * fmaxf(h[i], 0.f) + (1.f - s) * pred[i];
    }
}
__global__ void highway_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ h, const float* __restrict__ pred,
                                   float* __restrict__ dh, float* __restrict__ dpred, long n) {
    EW_LOOP(i, n) {
        const float hv = h[i], s = 1.f / (1.f + expf(-hv)), r = fmaxf(hv, 0.f), g = dout[i];
        dh[i] = g * (s * (1.f - s) * (r - pred[i]) + (hv > 0.f ? s : 0.f));
        dpred[i] = g * (1.f - s);
    }
}

extern "C" int cst_highway_fwd(const float* h, const float* pred, float* out, long n, void* stream) {
    CST_REQUIRE(h && pred && out && n > 0, "cst_highway_fwd: bad arguments");
    hipLaunchKernelGGL(highway_fwd_kernel, ew_grid(n), dim3(EW_THREADS), 0, (hipStream_t)stream, h, pred, out, n);
    CST_LAUNCH_CHECK("cst_highway_fwd");
    return CST_OK;
}
extern "C" int cst_highway_bwd(const float* dout, const float* h, const float* pred, float* dh, float* dpred, long n, void* stream) {
    CST_REQUIRE(dout && h && pred && dh && dpred && n > 0, "cst_highway_bwd: bad arguments");
    hipLaunchKernelGGL(highway_bwd_kernel, ew_grid(n), dim3(EW_THREADS), 0, (hipStream_t)stream, dout, h, pred, dh, dpred, n);
    CST_LAUNCH_CHECK("cst_highway_bwd");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// scalar losses over small vectors (n <= a few thousand): one block, deterministic
//   kind 0: MSE vs target vector t (or constant tconst when t == null)
//   kind 1: BCE-with-logits vs constant tconst
// loss[0] = weight * mean(...);  dx = gscale * d(mean)/dx
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void small_loss_kernel(const float* __restrict__ x, const float* __restrict__ t, float tconst,
                                                          int kind, long n, float weight, float* __restrict__ loss,
                                                          float* __restrict__ dx, float gscale) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) {
        const float xv = x[i], tv = t ? t[i] : tconst;
        if (kind == 0) {
            const float df = xv - tv;
            s += df * df;
            if (dx) dx[i] = gscale * 2.f * df / (float)n;
        } else {
            s += fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv)));
            if (dx) dx[i] = gscale * (1.f / (1.f + expf(-xv)) - tv) / (float)n;
        }
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) loss[0] = weight * s / (float)n;
}

extern "C" int cst_small_loss(const float* x, const float* t, float tconst, int kind, long n, float weight, float* loss,
                              float* dx, float gscale, void* stream) {
    CST_REQUIRE(x && loss && n > 0 && (kind == 0 || kind == 1), "cst_small_loss: bad arguments");
    hipLaunchKernelGGL(small_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, t, tconst, kind, n, weight, loss, dx, gscale);
    CST_LAUNCH_CHECK("cst_small_loss");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// optimiser: sum of squares -> clip coefficient -> Adam, all on device (no host sync)
// ---------------------------------------------------------------------------------------------
// Deterministic by construction: block b writes its partial to partials[b], then ONE block adds the partials in index
// order (fixed tree).  A float atomicAdd per block would make the norm -- hence the clip coefficient, hence every
// parameter -- depend on block arrival order in the last bit, and data-parallel replicas that must stay bit-identical
// (parallel.check_replicas) would drift apart.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ partials) {
    __shared__ float red[16];
    float s = 0.f;
    EW_LOOP(i, n) s += g[i] * g[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* __restrict__ partials, int nb, float* __restrict__ out) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) s += partials[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) *out += s;
}

#define CST_SUMSQ_PARTIALS 1024
extern "C" int cst_sumsq_accumulate(const float* g, long n, float* out, float* partials, void* stream) {
    CST_REQUIRE(g && out && partials && n > 0, "cst_sumsq_accumulate: bad arguments");
    long b = (n + EW_THREADS * 8 - 1) / (EW_THREADS * 8); if (b > CST_SUMSQ_PARTIALS) b = CST_SUMSQ_PARTIALS; if (b < 1) b = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)b), dim3(256), 0, (hipStream_t)stream, g, n, partials);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, (int)b, out);
    CST_LAUNCH_CHECK("cst_sumsq_accumulate");
    return CST_OK;
}

// *out += sum of n partial sums of squares, in a fixed order (the partials cst_multi_accumulate left)
extern "C" int cst_sumsq_partials(const float* partials, int n, float* out, void* stream) {
    CST_REQUIRE(partials && out && n > 0, "cst_sumsq_partials: bad arguments");
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, n, out);
    CST_LAUNCH_CHECK("cst_sumsq_partials");
    return CST_OK;
}

// torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), applied only when < 1
__global__ void clip_scale_kernel(float* __restrict__ g, long n, const float* __restrict__ sumsq, float max_norm) {
    const float norm = sqrtf(*sumsq);
    const float coef = max_norm / (norm + 1e-6f);
    if (coef >= 1.f) return;
    EW_LOOP(i, n) g[i] *= coef;
}

extern "C" int cst_clip_scale(float* g, long n, const float* sumsq_dev, float max_norm, void* stream) {
    CST_REQUIRE(g && sumsq_dev && n > 0, "cst_clip_scale: bad arguments");
    hipLaunchKernelGGL(clip_scale_kernel, ew_grid(n), dim3(EW_THREADS), 0, (hipStream_t)stream, g, n, sumsq_dev, max_norm);
    CST_LAUNCH_CHECK("cst_clip_scale");
    return CST_OK;
}

// torch.optim.Adam (no weight decay, no amsgrad).  step_dev holds the 1-based step count.
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr, float b1, float b2, float eps, const int* __restrict__ step_dev) {
    const float t = (float)(*step_dev);
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1, rbc2 = 1.f / sqrtf(bc2);
    EW_LOOP(i, n) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * rbc2 + eps);
    }
}

// The same step with the global-norm clip folded in: g is read as g * min(1, max_norm / (sqrt(*sumsq) + 1e-6)), exactly the value
// cst_clip_scale would have stored (same fp32 product), without the extra read-modify-write pass over the gradients.
__global__ void adam_clipped_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                    long n, float lr, float b1, float b2, float eps, const int* __restrict__ step_dev,
                                    const float* __restrict__ sumsq, float max_norm) {
    const float t = (float)(*step_dev);
    const float bc1 = 1.f - powf(b1, t), bc2 = 1.f - powf(b2, t);
    const float step_size = lr / bc1, rbc2 = 1.f / sqrtf(bc2);
    const float norm = sqrtf(*sumsq);
    const float coef = max_norm / (norm + 1e-6f);
    const bool scale = !(coef >= 1.f);         // as clip_scale_kernel: a NaN norm gives a NaN coefficient that IS applied (torch.nn.utils.clip_grad_norm_ does the same)
    EW_LOOP(i, n) {
        const float g0 = g[i];
        const float gi = scale ? g0 * coef : g0;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) * rbc2 + eps);
    }
}

extern "C" int cst_adam_step_clipped(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                                     const int* step_dev, const float* sumsq_dev, float max_norm, void* stream) {
    CST_REQUIRE(p && g && m && v && step_dev && sumsq_dev && n > 0 && max_norm > 0.f, "cst_adam_step_clipped: bad arguments");
    hipLaunchKernelGGL(adam_clipped_kernel, ew_grid(n), dim3(EW_THREADS), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, step_dev,
                       sumsq_dev, max_norm);
    CST_LAUNCH_CHECK("cst_adam_step_clipped");
    return CST_OK;
}

extern "C" int cst_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                             const int* step_dev, void* stream) {
    CST_REQUIRE(p && g && m && v && step_dev && n > 0, "cst_adam_step: bad arguments");
    hipLaunchKernelGGL(adam_kernel, ew_grid(n), dim3(EW_THREADS), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps, step_dev);
    CST_LAUNCH_CHECK("cst_adam_step");
    return CST_OK;
}

// zero fill as a kernel (see cst_common.h): 16-byte stores where the pointer allows, words otherwise
__global__ __launch_bounds__(256) void zero_words_kernel(uint32_t* __restrict__ p, long n) {
    const long n4 = ((reinterpret_cast<uintptr_t>(p) & 15) == 0) ? (n >> 2) : 0;
    uint4* p4 = reinterpret_cast<uint4*>(p);
    EW_LOOP(i, n4) p4[i] = make_uint4(0u, 0u, 0u, 0u);
    for (long i = 4 * n4 + (long)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (long)gridDim.x * EW_THREADS) p[i] = 0u;
}
int cst_zero_words(void* p, long n_words, hipStream_t st) {
    if (n_words <= 0) return CST_OK;
    long b = (n_words / 4 + EW_THREADS - 1) / EW_THREADS; if (b > 2048) b = 2048; if (b < 1) b = 1;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)b), dim3(EW_THREADS), 0, st, (uint32_t*)p, n_words);
    return hipGetLastError() == hipSuccess ? CST_OK : CST_ERR_LAUNCH;
}
extern "C" int cst_zero(void* p, long n_bytes, void* stream) {
    CST_REQUIRE(p && n_bytes >= 0 && n_bytes % 4 == 0 && ((uintptr_t)p & 3) == 0, "cst_zero: pointer and size must be 4-byte aligned");
    if (cst_zero_words(p, n_bytes / 4, (hipStream_t)stream) != CST_OK) { cst_set_error("cst_zero: launch failed"); return CST_ERR_LAUNCH; }
    return CST_OK;
}

// counters that live on the device so a captured graph advances them on every replay
__global__ void add_i32_kernel(int* p, int inc) { if (threadIdx.x == 0 && blockIdx.x == 0) *p += inc; }
extern "C" int cst_add_i32(int* p, int inc, void* stream) {
    CST_REQUIRE(p, "cst_add_i32: null pointer");
    hipLaunchKernelGGL(add_i32_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, inc);
    CST_LAUNCH_CHECK("cst_add_i32");
    return CST_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-tensor gather: flat[dst_off[t] + i] (+)= src_t[i] for every tensor t in one launch.
// Chunk c covers elements [chunk_start[c], chunk_start[c] + CHUNK) of tensor chunk_tensor[c].
// Moves the per-parameter gradients autograd produced into the flat gradient buffer that the
// clip / Adam kernels and the RCCL all-reduce work on.
// ---------------------------------------------------------------------------------------------
#define MT_CHUNK 4096
// ssq (optional, [gridDim.x]): the sum of squares of what the chunk's slot of `flat` holds afterwards -- the global-norm clip that follows
// (clip_grad_norm_) then needs no pass of its own over the gradients, only the fixed-order sum of these partials (cst_sumsq_partials).
__global__ __launch_bounds__(256) void multi_accumulate_kernel(const float* const* __restrict__ srcs, const long* __restrict__ dst_off,
                                                               const long* __restrict__ sizes, const int* __restrict__ chunk_tensor,
                                                               const long* __restrict__ chunk_start, float* __restrict__ flat,
                                                               int accumulate, float* __restrict__ ssq) {
    __shared__ float red[16];
    const int t = chunk_tensor[blockIdx.x];
    const float* src = srcs[t];
    if (!src && !ssq) return;
    const long s0 = chunk_start[blockIdx.x], n = sizes[t];
    float* dst = flat + dst_off[t];
    const long end = min(n, s0 + MT_CHUNK);
    constexpr int NIT = MT_CHUNK / 256;                        // 16 elements per thread: all loads issued before the first store
    float v[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const long i = min(s0 + threadIdx.x + 256L * k, end - 1);
        v[k] = src ? src[i] : dst[i];                           // no gradient for this tensor: the slot keeps what it holds (zero, or what accumulated)
        if (src && accumulate) v[k] += dst[i];
    }
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const long i = s0 + threadIdx.x + 256L * k;
        if (i < end) {
            if (src) dst[i] = v[k];
            q += v[k] * v[k];
        }
    }
    if (ssq) {
        q = block_sum(q, red);
        if (threadIdx.x == 0) ssq[blockIdx.x] = q;
    }
}

extern "C" int cst_multi_accumulate(const void* srcs_dev, const long* dst_off_dev, const long* sizes_dev,
                                    const int* chunk_tensor_dev, const long* chunk_start_dev, int nchunks,
                                    float* flat, int accumulate, float* ssq_partials, void* stream) {
    CST_REQUIRE(srcs_dev && dst_off_dev && sizes_dev && chunk_tensor_dev && chunk_start_dev && flat && nchunks > 0,
                "cst_multi_accumulate: bad arguments");
    hipLaunchKernelGGL(multi_accumulate_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream,
                       (const float* const*)srcs_dev, dst_off_dev, sizes_dev, chunk_tensor_dev, chunk_start_dev, flat, accumulate, ssq_partials);
    CST_LAUNCH_CHECK("cst_multi_accumulate");
    return CST_OK;
}
