/* libcst_hip.so -- C ABI of the MI355X (gfx950) kernels behind the three-stage style-transfer
 * training path (reference: src/main_pretrain.py -> main_warmup.py -> main_optimize.py and
 * src/model/{rnn,mlm,match,classifier,discriminator}.py of iptmt/consistent__style_transfer).
 *
 * The reference has no FFI of its own: its L0 is torch ops called from src/model/*.py
 * (SURVEY.md section 1).  This header is the boundary the build introduces beneath that Python
 * module API; each entry point names the reference call sites whose arithmetic it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer (hipMalloc'ed memory,
 *     e.g. torch tensor.data_ptr()) unless its name ends in _host;
 *   - matrices are fp32 row-major with an explicit leading dimension in ELEMENTS; token ids
 *     are int64 (loader.py:63-68 builds torch.long tensors);
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous and ordered only by
 *     that stream; nothing synchronises, allocates or frees (graph-capture safe);
 *   - the caller owns every buffer; workspace sizes come from the *_workspace_floats twins (every entry point that takes a
 *     workspace has one);
 *   - return value: 0 ok, 1 argument error, 2 launch error; cst_last_error() gives the text;
 *     nothing throws across the boundary;
 *   - dropout: (p, seed, stream_id, seed_dev) -- keep(idx) <=> (mix32(seed + *seed_dev,
 *     stream_id, idx) >> 8) >= floor(p * 2^24), idx = row-major element index of the tensor the
 *     mask applies to (oracle/rng.py is the same integer arithmetic); p = 0 disables;
 *   - one process per GPU; re-entrant across processes, not across threads on one stream.
 */
#ifndef CST_HIP_H
#define CST_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

const char* cst_last_error(void);
int cst_abi_version(void);

/* Data-parallel dropout contract (new work: the reference has no distributed code).  Every tensor a
 * dropout mask applies to is batch-major, so the element the one-process global batch indexes as idx sits at
 * rank * numel_local + idx_local on the rank that holds its batch row.  After cst_set_drop_shard(rank) every kernel of this
 * process adds rank * (elements of its local mask tensor) to its element indices: a shard draws exactly the masks the
 * one-process run draws for the same sentences (rnn.py:40,59,79,96; classifier.py:20,37; discriminator.py:29,48; the dropouts
 * inside nn.TransformerEncoderLayer, mlm.py:20-22 / match.py:18-20).  rank 0 (the default) = unsharded.  Process-global. */
int cst_set_drop_shard(int rank);

/* C[M,N] = epilogue(alpha * op(A)[M,K] . op(B)[K,N]) on the matrix cores.
 * a_kmajor: A is [M,lda] (1) or [K,lda] (0);  b_kmajor: B is [N,ldb] (1, a torch Linear weight
 * used as in F.linear) or [K,ldb] (0).  Epilogue order: +bias[n], +addend[m,n], act
 * (0 none, 1 relu, 2 LeakyReLU(0.1), 3 aux>0 ? v*gate_scale : 0, 4 aux>0 ? v : 0.1 v),
 * dropout over index m*N+n, optional C += v.  precision_f32 = 1 uses v_mfma_f32_16x16x4_f32
 * (exact fp32), 0 uses v_mfma_f32_16x16x32_bf16 with fp32 accumulation.  batch > 1 repeats with
 * the element strides s*.  tile: 0 auto, 64 or 128.  splitk: 0 auto (few output tiles and a long
 * K are cut into K-slices whose partial sums go to `workspace` and are summed in slice order by a
 * second pass that applies the epilogue), 1 never, >1 forced; workspace_floats >= batch*splits*M*N.
 * Replaces: nn.Linear / F.linear / tensor.matmul at rnn.py:35-38,61,79-80,85; mlm.py:24,31,45;
 * match.py:22,28,43; classifier.py:21,27,37; discriminator.py:28,35-37,39,45,48-49; the packed
 * in_proj / out_proj / linear1 / linear2 of nn.TransformerEncoderLayer (mlm.py:20-22,
 * match.py:18-20); nn.LSTM's gate projections (rnn.py:25-33); conv-as-GEMM for classifier.py:18
 * and discriminator.py:21-24; and all their autograd backward products. */
int cst_gemm(const float* A, long lda, int a_kmajor, const float* B, long ldb, int b_kmajor,
             float* C, long ldc, int M, int N, int K,
             const float* bias, const float* addend, long ldadd,
             const float* aux, long ldaux, int act, float gate_scale,
             int accumulate, float alpha, int precision_f32,
             int batch, long sA, long sB, long sC, long sBias, long sAdd, long sAux,
             float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
             int tile, int splitk, float* workspace, long workspace_floats, void* stream);
/* Floats of split-K workspace cst_gemm would use for this problem when given an unlimited one (0: single pass).  A smaller
 * workspace is never an error: the library lowers the split count to what fits. */
long cst_gemm_workspace_floats(int M, int N, int K, int batch, int precision_f32, int tile, int splitk);

/* Kernel-precise timing of the two GEMM kernels for roofline reporting: while enabled every launch
 * carries a start and a stop HIP event bound to the dispatch (hipExtLaunchKernelGGL); the read call
 * returns the summed kernel time, FLOPs (2MNK) and minimal operand bytes of kernel `which`
 * (0 = cst_gemm_kernel, 1 = cst_gemm_bf16_kernel; read 0 first, reading 1 clears).  Not capture-safe. */
int cst_gemm_profile_enable(int on);
int cst_gemm_profile_read(int which, double* total_ms_host, double* total_flops_host, double* total_min_bytes_host, long* launches_host);
/* Per-launch records of the same list (host arrays of max_records entries): M, N, K of every GEMM launch (K < 0: the
 * transposed-read weight-gradient product, contraction over rows), its duration and kernel family -- bench.py's
 * roofline.by_shape.  Call before cst_gemm_profile_read(1, ...), which clears the list. */
int cst_gemm_profile_shapes(long max_records, int* mnk_host, double* ms_host, int* which_host, long* count_host);

/* bf16-operand NT GEMM with direct-to-LDS (global_load_lds) staging in a 2-stage ring (`tile` 0 = chosen by shape between
 * 64x128 and 128x128; 64 / 128 force one; 65 / 129 / 130 = deeper rings, 256 / 252 / 248 = the 256-wide and loader/consumer
 * kernels, measured slower and kept for benchmarks only -- they need M % 256, N % 256 (248: N % 128) and no split-K):
 * C[M,N] (fp32) and/or Cb[M,N] (bf16) = epilogue(alpha * A[M,K] . B[N,K]^T); A, B bf16 (uint16 storage),
 * K contiguous and a multiple of 64 (zero padding is written by cst_cast_bf16), lda/ldb multiples of
 * 8, 16-byte aligned.  Epilogue as cst_gemm (aux is bf16).  The encoder layers' QKV / out-proj / FFN
 * products (mlm.py:20-22, match.py:18-20) run on it in bf16 mode: forward as is, dgrad with the
 * transposed weight copy, wgrad with the transposed activation copies -- all NT. */
int cst_gemm_bf16(const void* A, long lda, const void* B, long ldb,
                  float* C, long ldc, void* Cb, long ldcb, int M, int N, int K,
                  const float* bias, const float* addend, long ldadd, const void* aux, long ldaux,
                  int act, float gate_scale, float alpha, int accumulate,
                  float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                  int tile, int splitk, float* workspace, long workspace_floats, void* stream);
/* The same query for cst_gemm_bf16 (and cst_gemm_bf16_w8: pass tile = 64). */
long cst_gemm_bf16_workspace_floats(int M, int N, int K, int tile, int splitk);
/* Round 4: which kernel cst_gemm_bf16 (tile = 0, no split asked for) runs a product of this shape on.  100 TM + TN > 0: the big-tile
 * ping-pong kernel of csrc/gemm_pp.hip with workgroup tiles of (32 TM) x (64 TN) -- eight waves in two groups half a phase apart, one
 * persistent workgroup per CU, the tile shape chosen so that the tile count fills whole rounds of the CUs; 0: the LDS-DMA tile kernels
 * (64 x 128 / 128 x 128, split-K through the workspace).  Tile code 1000 + 100 TM + TN forces a wave tile, any other non-zero tile code
 * (64, 128, 136, 999) keeps the product on the tile kernels.  Same results up to fp32 summation order (one K order per output element). */
int cst_gemm_bf16_pp_config(int M, int N, int K);
/* out[r, 0..ldo) = bf16(x[r,:] * dropmask) zero-padded, out_t[c, 0..ldot) = the transpose zero-padded
 * (either may be null); x is fp32, or bf16 when x_is_bf16 (pure transpose). */
int cst_cast_bf16(const void* x, int x_is_bf16, long ldx, int R, int C,
                  void* out, long ldo, void* out_t, long ldot,
                  float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);
/* n cst_cast_bf16 calls (fp32 inputs, no dropout) in ONE launch: the bf16 twins (row-major + transposed) of every trained weight
 * of an optimizer group, refreshed once per optimizer step instead of one launch per weight (mlm.py:20-24, match.py:18-22 Linear
 * weights; the casts were 84 launches of the step).  table: n rows of 8 HOST int64 words {x, ldx, R, C, out, ldo, out_t, ldot},
 * the arguments cst_cast_bf16 takes; every row must qualify for its vector kernel (C, ldx, ldo, ldot multiples of 4, x 16-byte and
 * the outputs 8-byte aligned).  The rows ride in the kernel arguments (batches of 40): nothing is read from `table` after return. */
int cst_cast_bf16_multi(const long* table, int n, void* stream);
/* out[c] (+)= sum_r X[r,c] for a bf16 matrix (fp32 sums); accumulate != 0: `out` already holds the value the sums are added to
 * (e.g. zeros from the caller's own arena -- the library then skips its fill kernel). */
int cst_colsum_bf16(const void* X, long ld, int M, int N, float* out, int accumulate, void* stream);

/* Fused token cross-entropy forward + backward: row_loss[r] = logsumexp(x_r) - x_r[target_r];
 * dlogits = grad_scale * (softmax(x_r) - onehot(target_r)) (may alias logits; null = forward
 * only).  Rows with a target outside [0,V) contribute 0.  Replaces nn.CrossEntropyLoss at
 * main_pretrain.py:71,73; main_warmup.py:52; main_optimize.py:106,109,137,139 (the mean is
 * cst_reduce_sum with scale 1/R -- PAD rows count, ignore_index stays at its default). */
int cst_token_ce(const float* logits, long ld, const int64_t* target, int R, int V,
                 float* row_loss, float* dlogits, long ldd, float grad_scale, void* stream);
/* _b: dlogits also in bf16 [R, lddb] with zeros in columns V..lddb-1 (K padding): the operand of the vocabulary
 * projection's dgrad and weight-gradient GEMMs. */
int cst_token_ce_b(const float* logits, long ld, const int64_t* target, int R, int V,
                   float* row_loss, float* dlogits, long ldd, float grad_scale,
                   void* dlogits_bf16, long lddb, void* stream);

/* p = softmax(logits * inv_tau) over V, argmax_out[r] = first index of max(p) (may be null).
 * rnn.py:83 (softmax(logits / tau)) and the argmax inside hard_sample, rnn.py:52-53. */
int cst_softmax_tau(const float* logits, long ld, float inv_tau, float* p, long ldp,
                    int64_t* argmax_out, int R, int V, void* stream);
/* The same, and for each row the embedding fed to the next decode step without a launch of its own:
 * gout[r, :E] = table[sel(r)] * dropout, sel(r) = (*coin_dev != 0 or ids_b == NULL) ? argmax(r) : ids_b[r*ldb]
 * (rnn.py:88-96; arguments as cst_embed_gather, dropout index r*E + c). */
int cst_softmax_tau_gather(const float* logits, long ld, float inv_tau, float* p, long ldp,
                           int64_t* argmax_out, int R, int V,
                           const float* table, long ldt, int E, float* out, long ldo, void* out_bf16, long ldob,
                           const int64_t* ids_b, long ldb, const int* coin_dev,
                           float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                           void* stream);
/* The same, also writing the bf16 twin of the probabilities (row stride ldpb, zero in columns [V, wpb), wpb < V + 64):
 * the operand of the products that embed all steps' distributions at once (main_optimize.py:96-111: matcher,
 * classifier and discriminator embed sample_p).  table == NULL: no fed-back embedding (last decode step). */
int cst_softmax_tau_gather_b(const float* logits, long ld, float inv_tau, float* p, long ldp,
                             void* p_bf16, long ldpb, int wpb, int64_t* argmax_out, int R, int V,
                             const float* table, long ldt, int E, float* out, long ldo, void* out_bf16, long ldob,
                             const int64_t* ids_b, long ldb, const int* coin_dev,
                             float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                             void* stream);
/* dx = inv_tau * p * (dp - sum(dp * p)); dx may alias dp. */
int cst_softmax_tau_bwd(const float* p, long ldp, const float* dp, long lddp, float inv_tau,
                        float* dx, long lddx, void* dx_bf16, long lddxb, int R, int V, void* stream);
/* out[r] = first index of the row maximum (rnn.py:92; main_optimize.py:104,131,162). */
int cst_argmax_rows(const float* x, long ld, int R, int V, int64_t* out, void* stream);
int cst_argmax_rows_gather(const float* x, long ld, int R, int V, int64_t* out,
                           const float* table, long ldt, int E, float* gout, long ldo, void* out_bf16, long ldob,
                           const int64_t* ids_b, long ldb, const int* coin_dev,
                           float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                           void* stream);

/* z = res + dropout(x); y = LayerNorm(z) (eps inside the sqrt, biased variance); z may alias x or
 * be null.  norm1/norm2 of nn.TransformerEncoderLayer with its dropout1/dropout2 and residuals. */
int cst_add_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                          float* z, float* y, float* mean, float* rstd, int T, int d,
                          float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                          void* stream);
/* _b: y also in bf16 (leading dimension ldyb): the A operand of the next projection. */
int cst_add_layernorm_fwd_b(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                            float* z, float* y, float* mean, float* rstd, int T, int d,
                            float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                            void* y_bf16, long ldyb, void* stream);
long cst_layernorm_bwd_workspace_floats(int T, int d);
int cst_layernorm_bwd(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                      float* dz, float* dgamma, float* dbeta, int accumulate,
                      float* workspace, long workspace_floats, int T, int d, void* stream);
/* _b: also dz_bf16 = bf16(dropout'(dz)) with the given dropout descriptor (element index row*d + c): the gradient that
 * flows through the dropout in front of the residual add, as the GEMMs behind this LayerNorm consume it.
 * dparams3 (optional, [3d]): (dgamma | dbeta | column sums of dropout'(dz) = the bias gradient of the Linear in front of the
 * LayerNorm) finished by ONE column-sum pass; needs 3*nblk*d workspace floats (1.5x cst_layernorm_bwd_workspace_floats). */
int cst_layernorm_bwd_b(const float* dy, const float* z, const float* mean, const float* rstd, const float* gamma,
                        float* dz, float* dgamma, float* dbeta, int accumulate,
                        float* workspace, long workspace_floats, int T, int d,
                        void* dz_bf16, long lddzb, float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                        float* dparams3, void* stream);
/* Zero n_bytes (multiple of 4) at p with a kernel launch: the zero-initialised gradient accumulators of the backward passes
 * (embedding scatter targets, d memory, ...).  A kernel, not hipMemsetAsync: memset nodes issued by the autograd thread lose
 * their place in the stream order under segmented hipGraph capture (thread_local capture mode next to RCCL). */
int cst_zero(void* p, long n_bytes, void* stream);
/* out[c] (+)= sum_r X[r,c]  (bias gradients). */
int cst_colsum(const float* X, long ld, int M, int N, float* out, int accumulate, void* stream);
/* out[0] (+)= scale * sum(in[0..n)), one block, deterministic. */
int cst_reduce_sum(const float* in, long n, float scale, float* out, int accumulate, void* stream);

/* Unmasked multi-head self-attention core, qkv [B,S,3d] -> out [B,S,d], lse [B,H,S]; attention dropout on the
 * probabilities (index ((b*H+h)*S+i)*S+j).  S <= 64: one LDS-resident S x S plane per (batch row, head), head dim in
 * {8,16,32,64,96}; 64 < S <= 128 (the book corpus: match.py:36-39 concatenates two sequences of max_len 30,
 * arguments.py:43, lengthened by data_util.transfer_noise): K/V resident, queries in blocks, head dim in {8,64,96}.
 * nn.MultiheadAttention inside nn.TransformerEncoderLayer (mlm.py:20-22,43; match.py:18-20,39). */
int cst_mha_fwd(const float* qkv, float* out, float* lse, int B, int S, int H, int hd,
                float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);
/* _b: also writes the result in bf16 (row-major, leading dimension ldob / lddb elements) -- the A operand of the
 * projection GEMM that consumes it, saving a separate cast pass. */
int cst_mha_fwd_b(const float* qkv, float* out, float* lse, int B, int S, int H, int hd,
                  float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                  void* out_bf16, long ldob, void* stream);
/* _h: the same attention core with bf16 qkv [B,S,3d] and (backward) bf16 d(attention output) [B,S,d] in HBM and optional
 * fp32 results (out / dqkv may be null when the bf16 twin is all the caller consumes): in bf16 mode the encoder layer keeps
 * qkv, the attention output, its gradient and dqkv in bf16 only, which halves-to-thirds the attention core's HBM traffic.
 * S <= 64, head dims 64 / 96.  The LDS images of q, k, v, d(out) are bf16 (two to four workgroups per CU); Q K^T and dO V^T run on
 * v_mfma_f32_16x16x32_bf16 (exact products of the bf16 inputs), softmax and every other product in fp32: results equal cst_mha_fwd_b /
 * cst_mha_bwd_b run on the same (bf16-rounded) values up to summation order. */
int cst_mha_fwd_h(const void* qkv_bf16, float* out, float* lse, int B, int S, int H, int hd,
                  float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                  void* out_bf16, long ldob, void* stream);
int cst_mha_bwd_h(const void* qkv_bf16, const void* dout_bf16, const float* lse, float* dqkv, int B, int S, int H, int hd,
                  float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                  void* dqkv_bf16, long lddb, void* stream);
int cst_mha_bwd(const float* qkv, const float* dout, const float* lse, float* dqkv, int B, int S, int H, int hd,
                float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);
int cst_mha_bwd_b(const float* qkv, const float* dout, const float* lse, float* dqkv, int B, int S, int H, int hd,
                  float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                  void* dqkv_bf16, long lddb, void* stream);

/* Single-query dot attention (rnn.py:46-50,76): out = softmax(q mem^T / sqrt(D)) mem; p [B,L] kept.
 * `dropped` (optional, [B, lddrop]): also writes dropout([q | out]) -- the decoder's i_ffn of
 * rnn.py:78-79, dropout index b*2D + c -- saving a separate dropout launch per decode step. */
int cst_dot_attn_fwd(const float* q, long ldq, const float* mem, float* out, long ldo, float* p,
                     int B, int L, int D, float* dropped, long lddrop, void* dropped_bf16, long lddropb,
                     float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);
/* dq (+)= ..., dmem += ... (dmem accumulates across decode steps; zero it first).
 * _steps: all T decode steps of a batch row in one workgroup, for decodes whose FFN gradients are known up front:
 * diffn [B, T, 2D] holds d[h | a] of the FFN input before its dropout (stream drop_stream + s, element b*2D + c);
 * on return diffn[b, s, 0:D] = dropout(.)[0:D] + dq.  q [B, T, .]: h_s rows; p [T, B, L]; dmem += sum over steps. */
int cst_dot_attn_bwd(const float* dout, long lddo, const float* q, long ldq, const float* mem, const float* p,
                     float* dq, long lddq, int dq_accumulate, float* dmem, int B, int L, int D, void* stream);
int cst_dot_attn_bwd_steps(float* diffn, long ldrow, long gstep, const float* q, long ldq, long qstep,
                           const float* mem, const float* p, float* dmem, int B, int T, int L, int D,
                           float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);

/* LSTM cell (gate order i,f,g,o as nn.LSTM, rnn.py:25-33): gates [B,4H] pre-activation in,
 * activations out (kept for backward); h_out2 optional second copy of h; h_bf16 / h_bf16_2 optional
 * bf16 copies (A operands of the next GEMMs); cell_bwd: dgates_bf16 likewise. */
int cst_lstm_cell_fwd(float* gates, long ldg, const float* c_prev, long ldcp, float* h_out, long ldh,
                      float* c_out, long ldc, float* h_out2, long ldh2,
                      void* h_bf16, long ldhb, void* h_bf16_2, long ldhb2, int B, int H, void* stream);
/* W8A16 (BASELINE configs[4], "fp8 weight MFMA + bf16 activations"): the weights of the encoder layers' Linear products
 * (packed in_proj, out_proj, linear1, linear2 of nn.TransformerEncoderLayer: mlm.py:20-22, match.py:18-20) as fp8 e4m3 (OCP)
 * with one fp32 scale per output channel.
 * cst_cast_fp8_rows: out[r, c] = fp8(W(r, c) / scale[r]), scale[r] = max_c |W(r, c)| / 448; W(r, c) = W[r * ldw + c * colstride]
 *   (colstride > 1 quantises a transposed view: the dgrad product needs W^T with scales along its own rows); out is [R, ldo]
 *   bytes, ldo a multiple of 16 >= C, columns >= C zero.
 * cst_gemm_bf16_w8: C / Cb [M,N] = epi(alpha * bscale[n] * A[M,K] Bq[N,K]^T), A bf16 as in cst_gemm_bf16, Bq fp8 with ldb in
 *   bytes; the weight tile moves HBM -> LDS at one byte per element and is widened to bf16 in registers in front of
 *   v_mfma_f32_16x16x32_bf16 (gfx950 has no MFMA that mixes bf16 and fp8 operands); epilogues as cst_gemm_bf16. */
int cst_cast_fp8_rows(const float* W, long ldw, long colstride, int R, int C, void* out, long ldo, float* scale, void* stream);
int cst_gemm_bf16_w8(const void* A, long lda, const void* Bq, long ldb, const float* bscale,
                     float* C, long ldc, void* Cb, long ldcb, int M, int N, int K,
                     const float* bias, const float* addend, long ldadd, const void* aux, long ldaux,
                     int act, float gate_scale, float alpha, int accumulate,
                     float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                     int splitk, float* workspace, long workspace_floats, void* stream);
/* C[M,N] (+)= A^T B with A [K,M] and B [K,N] bf16, the CONTRACTION index being the row index of both: the weight
 * gradients dW = dY^T X of every Linear on the path (backward of mlm.py:20-24, match.py:18-22) straight from the
 * row-major bf16 activations, no transposed copies.  M, N multiples of 8, K a multiple of 64 (token count);
 * split-K as cst_gemm_bf16. */
int cst_gemm_bf16_tt(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                     int accumulate, int splitk, float* workspace, long workspace_floats, void* stream);
long cst_gemm_bf16_tt_workspace_floats(int M, int N, int K, int splitk);
/* Grouped weight gradients.  Between _group_begin and _group_end (one group per process at a time; begin and end may come from
 * different host threads, e.g. autograd's device thread and the thread that called backward) cst_gemm_bf16_tt calls with splitk 0 or 1
 * only RECORD their product (splitk < 0: run now with the automatic split; splitk >= 2: run now with that split); _group_end launches all of them as ONE kernel whose workgroups are numbered through the problems' 128 x 128
 * output tiles (up to 8 problems per launch; more are launched in batches of 8): the four dW of an encoder layer (backward of
 * mlm.py:20-24, match.py:18-22) have 36-108 tiles each -- one by one they need split-K slabs and a reduce launch each to fill 256
 * CUs, together (336 tiles) they do not.  A group of at most 256 tiles (d = 512 layers: 192) splits every contraction in two: two
 * workgroups per tile write partial tiles to the recorded calls' workspace and the one that finishes last (one counter per tile,
 * nobody waits) adds them in split order -- no second launch, the same sum whichever workgroup that is.
 *   counters: ncounters ints, ZERO when the group is launched; the kernel leaves them zero.  null: whole-K workgroups only.
 * A group with fewer than 128 tiles in total (env CST_TT_GROUP_MIN overrides the rule; 0 = never group) is launched product by
 * product exactly as without a group.  Operands and outputs must stay alive and unchanged
 * until _group_end returns; results are defined only after it.  _group_end always closes the group. */
int cst_gemm_bf16_tt_group_begin(int* counters, int ncounters, void* stream);
int cst_gemm_bf16_tt_group_end(void* stream);
/* diagnostics: how the last group was launched -- 0 product by product, S >= 1 one kernel with S workgroups per tile */
int cst_gemm_bf16_tt_group_last_splits(void);

/* All L time steps of both directions of the BiLSTM encoder (rnn.py:25-27, called at rnn.py:57, :62) in ONE launch: a
 * workgroup takes 16 batch rows of one direction through the whole sequence (recurrences are independent across batch
 * rows), h W_hh^T on the bf16 matrix pipe with W_hh streamed from L2 each step, cell state in registers.
 *   whh{0,1}: W_hh of the forward / reverse direction in bf16 MFMA-fragment order [wave 4][gate 4][tile 4][k step 8][lane 64][8]
 *   (element [w][q][j][kk][16*lq + lr][e] = W_hh[q*H + 64w + 16j + lr][32kk + 8lq + e]: every 64-lane load reads one
 *   contiguous KiB); xp{0,1}: [B, L*4H] input projections
 *   incl. biases; h0 [B, ldh0]: forward direction's initial state at column 0, reverse at column H; outputs as the
 *   per-step path writes them: gates{0,1} [L,B,4H] (activated), cenc{0,1} [L,B,H], hprev{0,1} [B, L*H] (h entering each
 *   time index; hprev{0,1}_bf16: optional bf16 twins, both or neither -- the operand of the transposed-read dW_hh product),
 *   c_last [B, ldcl] (final cell states at columns 0 / H), mem [B, L*2H] and its bf16 twin.
 * H must be 256 and B a multiple of 16 (status 1 otherwise: use the per-step entry points). */
int cst_lstm_seq_fwd(const void* whh0, const void* whh1, const float* xp0, const float* xp1,
                     const float* h0, long ldh0, float* gates0, float* gates1, float* cenc0, float* cenc1,
                     float* hprev0, float* hprev1, void* hprev0_bf16, void* hprev1_bf16,
                     float* c_last, long ldcl, float* mem, void* mem_bf16,
                     int B, int L, int H, void* stream);

/* cst_lstm_seq_fwd with TWO workgroups per (direction, 16-row group), each keeping its half of W_hh on chip for the whole launch (LDS +
 * registers: nothing is streamed) and exchanging its half of h_t with its partner once per step through tagged 8-byte granules in `xchg`
 * (cdna_hip_programming.md Guideline 16, recipe R2; zeroed by the entry point in front of every launch; bounded spins: a timeout sets the
 * last word of the workspace instead of hanging).  Same arguments and results as cst_lstm_seq_fwd; B <= 1024 (all workgroups co-resident). */
long cst_lstm_seq_xchg_bytes(int B);
int cst_lstm_seq_fwd_split(const void* whh0, const void* whh1, const float* xp0, const float* xp1,
                           const float* h0, long ldh0, float* gates0, float* gates1, float* cenc0, float* cenc1,
                           float* hprev0, float* hprev1, void* hprev0_bf16, void* hprev1_bf16,
                           float* c_last, long ldcl, float* mem, void* mem_bf16,
                           int B, int L, int H, void* xchg, long xchg_bytes, void* stream);
/* Residency of the split kernel: workgroups it launches for batch B (one whole CU each: 136 KB of LDS, 512 registers a lane) and the CU
 * count of the current device.  cst_lstm_seq_fwd_split refuses (status 1, before anything is launched or captured) when the first exceeds
 * the second; the caller keeps a margin beside other resident kernels (gen_fn.split_enabled).  On a timeout the kernel sets the sticky
 * word AND writes NaN into the partner's half of h_t, so every later gate, encoder state and c_last of that row group is NaN: the loss
 * of the same step is NaN, never a number computed from stale hidden states (rnn.py:62's encoder has no such failure mode to mirror). */
int cst_lstm_seq_split_workgroups(int B);
int cst_lstm_seq_split_capacity(void);
/* TEST ONLY: cst_lstm_seq_fwd_split with the second workgroup of every pair missing -- what lost co-residency looks like.  Every launched
 * workgroup times out at its first exchange (bounded spin, ~0.2 s), sets the sticky word and poisons its outputs with NaN. */
int cst_lstm_seq_fwd_split_lone_half(const void* whh0, const void* whh1, const float* xp0, const float* xp1,
                                     const float* h0, long ldh0, float* gates0, float* gates1, float* cenc0, float* cenc1,
                                     float* hprev0, float* hprev1, void* hprev0_bf16, void* hprev1_bf16,
                                     float* c_last, long ldcl, float* mem, void* mem_bf16,
                                     int B, int L, int H, void* xchg, long xchg_bytes, void* stream);
/* Backward of cst_lstm_seq_fwd, also one launch: per step the cell backward in registers and dh_{prev} = dgates W_hh on
 * the bf16 matrix pipe.  wt{0,1}: W_hh^T in bf16 fragment order [wave 4][k step 32][tile 4][lane 64][8] (element
 * [w][kk][j][16*lq + lr][e] = W_hh[32kk + 8lq + e][64w + 16j + lr]); gates / cenc / c_last as the forward wrote them;
 * dc_last [B, lddcl]: gradient w.r.t. the final cell states (columns 0 / H); dmem [B, L*2H]: gradient w.r.t. the encoder
 * states; outputs dgates{0,1} [B, L*4H] (pre-activation gate gradients at column t*4H), optionally their bf16 twins
 * (both or neither: operands of the weight-gradient and input-gradient products) and dh0 [B, lddh0]. */
int cst_lstm_seq_bwd(const void* wt0, const void* wt1, const float* gates0, const float* gates1,
                     const float* cenc0, const float* cenc1, const float* c_last, long ldcl,
                     const float* dc_last, long lddcl, const float* dmem,
                     float* dgates0, float* dgates1, void* dgates0_bf16, void* dgates1_bf16,
                     float* dh0, long lddh0, int B, int L, int H, void* stream);

/* Soft decode backward, one step behind the fn_1 dgrad (rnn.py:46-50 attention, :75 LSTM cell): single-query attention backward and cell
 * backward of every batch row in one launch (D == 512, L <= 64).  g = d[h | a] of the (dropped) FFN input, row b at g + b * ldg; writes the
 * step's ds [B][L], dgates (fp32 + optional bf16) and dc_prev (may alias dc).  d memory is left to cst_dec_attn_dmem. */
int cst_dec_attn_cell_bwd(const float* g, long ldg, const float* mem, const float* p, float* ds_out, int B, int L, int D,
                          const float* gates, long ldgt, const float* c_prev, long ldcp, const float* c_new, long ldcn,
                          const float* dh2, long lddh2, const float* dc, long lddc,
                          float* dgates, long lddg, float* dc_prev, long lddcp, void* dgates_bf16, long lddgb, void* stream);
/* Straight-through gradient of the soft decode (rnn.py:84-85), one step: C[M, V] += dropout(g)[M, 128] . E_bf16[V, 128]^T; gx_out (optional)
 * receives dropout(g) in fp32 (the embedding scatter's operand).  Dropout over the (M, 128) index space, as cst_dropout.  K == 128. */
int cst_dec_dxe(const float* g, long ldg, float* gx_out, long ldgx, const void* E_bf16, long lde, float* C, long ldc,
                int M, int V, int K, float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);
/* dmem[b, j, :] += sum_s p[s, b, j] ga[b, s] + ds[s, b, j] h[b, s]: the d memory of all T steps after the loop (p, ds: [T][B][L]). */
int cst_dec_attn_dmem(const float* ga, long ldga, long ga_step, const float* h, long ldh, long h_step,
                      const float* p, const float* ds, float* dmem, int B, int T, int L, int D, void* stream);

/* One recurrent step of nn.LSTM (rnn.py:25-33, called at rnn.py:57 and :75) in two launches, for one problem
 * or for two independent problems of one shape (the *2 / *_p2 arguments; A2 == NULL: single) -- the two
 * directions of the bidirectional encoder share every dimension and leading dimension.
 *   forward:  gates = A B^T (+ bias) (+ addend) as cst_gemm_bf16 (A [M,K], B [4H,K] bf16, K % 64 == 0), then the
 *             cell of cst_lstm_cell_fwd.
 *   backward: dh = A B^T (A = dgates of the step after [M,4H] bf16, B = W_hh^T [H,4H] bf16, K = 4H) (+ dh_extra),
 *             then the cell backward of cst_lstm_cell_bwd for this step (dc_in NULL = 0).  n_extra > 0: B has n_extra
 *             leading rows whose products are stored as they are in extra_out [M, ldx] (the decoder's
 *             [W_ih | W_hh]^T: d x_t next to d h_{t-1}).
 * The product's split-K partials go to the workspace (>= problems*splits*M*N floats; splitk 0 = heuristic) and ONE
 * second kernel sums them and applies the cell: no pre-activation / dh round trip, no separate reduce and cell
 * launches.  Every fp32 row pointer must be 16-byte aligned (H % 4 == 0, leading dimensions multiples of 4). */
int cst_gemm_bf16_lstm(const void* A, long lda, const void* B, long ldb, int M, int H, int K,
                       const float* bias, const float* addend, long ldadd,
                       float* gates, long ldg, const float* c_prev, long ldcp,
                       float* h_out, long ldh, float* c_out, long ldc, float* h_out2, long ldh2,
                       void* h_bf16, long ldhb, void* h_bf16_2, long ldhb2,
                       const void* A2, const void* B2, const float* bias2, const float* addend2,
                       float* gates2, const float* c_prev2, float* h_out_2, float* c_out2, float* h_out2_2,
                       void* h_bf16_p2, void* h_bf16_2_p2,
                       int splitk, float* workspace, long workspace_floats, void* stream);
/* Workspace of the recurrent products (cst_gemm_bf16_lstm / _lstm_bwd / _lstm_attn), which always write their partial sums to it:
 * N = 4H gate columns (the K of _lstm_bwd's product for that entry point), problems = 2 when both encoder directions go in one
 * launch.  These entry points REQUIRE at least this much. */
long cst_gemm_bf16_lstm_workspace_floats(int M, int N, int K, int problems, int splitk);
/* The decoder's variant of the forward step (rnn.py:75-79): gates product (bias only), cell, then the single-query
 * attention of cst_dot_attn_fwd with h_t as the query (D = H) and the FFN-input dropout, all in the second launch
 * (one workgroup per batch row).  Arguments as cst_gemm_bf16_lstm / cst_dot_attn_fwd. */
int cst_gemm_bf16_lstm_attn(const void* A, long lda, const void* B, long ldb, int M, int H, int K,
                            const float* bias, float* gates, long ldg, const float* c_prev, long ldcp,
                            float* h_out, long ldh, float* c_out, long ldc, float* h_out2, long ldh2, void* h_bf16_2, long ldhb2,
                            const float* mem, int L, float* att_out, long ldo, float* p,
                            float* dropped, long lddrop, void* dropped_bf16, long lddropb,
                            float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                            int splitk, float* workspace, long workspace_floats, void* stream);
/* ---- decoder step in four launches (csrc/decode.hip; rnn.py:72-97) --------------------------------------------------
 * cst_dec_gates: one decode step's LSTM cell, input side included (rnn.py:74, :88-96).
 *   A [B, E + Hd] bf16 = [x_t | h_{t-1}].  With amax_prev != NULL the x columns are not read: x_t = dropout(table[token]) is
 *   built in the kernel, token = the previous step's arg-max (packed words amax_prev[g][b], see cst_gemm_bf16_argmax) or, when
 *   ids_teacher != NULL and (coin_dev == NULL or *coin_dev == 0), the teacher token ids_teacher[b * ldids] (scheduled sampling,
 *   rnn.py:91-94: the coin is a device word so a captured graph stays static); dropout (p, seed, stream) over the (B, E) index
 *   space (rnn.py:96); its bf16 copy goes to x_bf16_out (operand of the weight gradient; may be A's own x columns).
 *   W [4 Hd, E + Hd] bf16 = [W_ih | W_hh], bias [4 Hd] = b_ih + b_hh; gates (i, f, g, o ACTIVATED, what the backward pass reads),
 *   c_out, h_out fp32; h_bf16_next (may be NULL) = the h columns of the next step's A.  Whole K in LDS (cst_dec_gates_lds_bytes),
 *   32 rows x 16 hidden units per workgroup, no K split, no reduce launch.  Needs Hd % 16 == 0, E == 128. */
long cst_dec_gates_lds_bytes(int E, int Hd);
int cst_dec_gates(const void* A, long lda, const void* W, long ldw,
                  const void* amax_prev, const int64_t* ids_teacher, long ldids, const int* coin_dev,
                  const float* table, long ldtab, int V,
                  float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev,
                  void* x_bf16_out, long ldxb, const float* bias, const float* c_prev, long ldcp,
                  float* gates, long ldg, float* c_out, long ldc, float* h_out, long ldh, void* h_bf16_next, long ldhb,
                  int B, int E, int Hd, void* stream);
/* cst_dot_attn_fwd for the decode loop's shapes (D == 512, L <= 64): out = softmax(q mem^T / sqrt(D)) mem per batch row (rnn.py:46-50,
 * :76), p = the attention weights (kept for the backward pass), dropped_bf16 (may be NULL) = bf16(dropout([q | out])) over the (B, 2D)
 * index space = the A operand of fn_1 (rnn.py:79).  The row's encoder states go straight to registers in one burst. */
int cst_dec_attn(const float* q, long ldq, const float* mem, float* out, long ldo, float* p, int B, int L, int D,
                 void* dropped_bf16, long lddropb,
                 float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);
/* C / Cb [M, N] = act(A[M, K] . B[N, K]^T + bias), bf16 operands with K contiguous, act 0 none / 1 relu / 2 LeakyReLU(0.1): the
 * M = batch products of a decode step that are too small to split (fn_1, rnn.py:79-80): whole K in LDS (K <= 1280), 32 x 32
 * tiles, one launch; dropout over the (M, N) index space last (the dgrad through dropout(i_ffn), rnn.py:79).  N a multiple of 32, K of 64. */
int cst_gemm_bf16_skinny(const void* A, long lda, const void* B, long ldb, float* C, long ldc, void* Cb, long ldcb,
                         int M, int N, int K, const float* bias, int act,
                         float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);
/* C[M, N] = A . B^T as cst_gemm_bf16 without epilogue operands, plus the arg-max of every row.  amax_packed: G = cst_argmax_groups()
 * words of 8 bytes per row, group-major [G][M], zeroed by the caller; word (group of a 128-column tile = tile index mod G, row m)
 * receives max over the tile's columns of ((order-preserving bits of C[m, n]) << 32 | (0xFFFFFFFF - n)) by 64-bit atomic max -- the
 * largest value and, among equals, the FIRST column, in any arrival order (torch.argmax, rnn.py:53, :92); the max over a row's G words
 * is the row's arg-max.  (G words per row, not one: one word would put every atomic of eight rows on a single 64-byte line.)
 * fn_2 of a decode step (rnn.py:80): the next step's cst_dec_gates reads the words, the loop needs no softmax / arg-max launch. */
int cst_argmax_groups(void);
int cst_gemm_bf16_argmax(const void* A, long lda, const void* B, long ldb, float* C, long ldc, int M, int N, int K,
                         void* amax_packed, void* stream);
/* The same product and arg-max words for K == 512 (the decoder's fn_2: logits = r1 W2^T, rnn.py:80) with the r1 rows STATIONARY: a
 * workgroup keeps its 64 rows of A as MFMA fragments in registers and streams a contiguous range of B's rows (vocabulary columns)
 * through a four-buffer LDS ring fed by dedicated loader waves; one arg-max atomic per row and workgroup.  C [M, V] fp32 (ldc >= V). */
int cst_dec_fn2(const void* A, long lda, const void* W, long ldw, float* C, long ldc, int M, int V, int K,
                void* amax_packed, void* stream);
/* ids[s][i] = column index held by the packed arg-max words of row i of product s: packed = `steps` blocks [G][n]
 * (rnn.py:88-92: the ids fed back; main_optimize.py:104 sample_p.argmax) */
int cst_unpack_argmax(const void* packed, int64_t* ids, long n, int steps, void* stream);
/* 1 when the library was built with -DCST_BENCH_VARIANTS (bench-only GEMM variants and timing ablations), else 0 */
int cst_bench_variants(void);

int cst_gemm_bf16_lstm_bwd(const void* A, long lda, const void* B, long ldb, int M, int H, int K,
                           const float* gates, long ldg, const float* c_prev, long ldcp, const float* c_new, long ldcn,
                           const float* dh_extra, long lddh, const float* dc_in, long lddc,
                           float* dgates, long lddg, float* dc_prev, long lddcp, void* dgates_bf16, long lddgb,
                           const void* A2, const void* B2, const float* gates2, const float* c_prev2, const float* c_new2,
                           const float* dh_extra2, const float* dc_in2, float* dgates2, float* dc_prev2, void* dgates_bf16_2,
                           int n_extra, float* extra_out, float* extra_out2, long ldx,
                           int splitk, float* workspace, long workspace_floats, void* stream);
int cst_lstm_cell_bwd(const float* gates, long ldg, const float* c_prev, long ldcp, const float* c_new, long ldcn,
                      const float* dh, long lddh, const float* dh2, long lddh2, const float* dc, long lddc,
                      float* dgates, long lddg, float* dc_prev, long lddcp, void* dgates_bf16, long lddgb,
                      int B, int H, void* stream);

/* out[r,:] = table[id(r)] * dropmask, id(r) = (*coin_dev) ? ids_a[r] : ids_b[r*ldb] (either list may
 * be null); transposed reads table[c*ldt + id].  nn.Embedding at rnn.py:59,95; classifier.py:25;
 * the one-hot path of discriminator.py:39 (main_optimize.py:117); the scheduled-sampling choice
 * of rnn.py:91-95 is made on the device. */
int cst_embed_gather(const int64_t* ids_a, const int64_t* ids_b, long ldb, const int* coin_dev,
                     const float* table, long ldt, int transposed, float* out, long ldo,
                     void* out_bf16, long ldob, int R, int E, int V,
                     float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);
int cst_embed_scatter_add(const int64_t* ids_a, const int64_t* ids_b, long ldb, const int* coin_dev,
                          const float* dout, long ldo, float* dtable, long ldt, int transposed, int R, int E, int V,
                          float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);

/* The decoder's S per-step embedding gradients in one launch (backward of rnn.py:88-96): row s*B + b of dout belongs
 * to the token fed to step s+1 -- ids_a[s*B+b] if coins_dev[s] != 0 (or ids_b == NULL) else ids_b[b*ldb + s] -- and
 * carries the dropout mask of call-site stream drop_stream + s, element b*E + c: identical to S launches of
 * cst_embed_scatter_add with (ids_a + s*B, ids_b + s, coin = coins_dev + s, drop_stream + s). */
int cst_embed_scatter_add_steps(const int64_t* ids_a, const int64_t* ids_b, long ldb, const int* coins_dev,
                                const float* dout, long ldo, float* dtable, long ldt, int S, int B, int E, int V,
                                float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);

/* x[b, off+l, :] = (Etok[ids[b,l]] | pre[b,l,:]) + Epos[l] (+ seg_row)   (mlm.py:27-38; match.py:24-34). */
int cst_tps_embed_fwd(const int64_t* ids, const float* pre, const float* Etok, const float* Epos, const float* seg_row,
                      float* x, int B, int L, int d, int S, int off, int V, void* stream);
int cst_tps_embed_bwd(const float* dx, const int64_t* ids, float* dpre, float* dEtok, float* dEpos, float* dseg_row,
                      int B, int L, int d, int S, int off, int V, void* stream);

/* im2col for the two convolution stacks: mode 0 = TextCNN (classifier.py:18,30: k x E window,
 * zero padding k-1), mode 1 = RelGAN_D (discriminator.py:21-24,41: k x (E/R) window, stride E/R). */
int cst_im2col(const float* e, float* col, int B, int L, int E, int k, int mode, int R, void* stream);
/* the same rows written in bf16 (dense): operands of the bf16 GEMMs of the convolution, no fp32 plane (bf16 mode, k E a multiple of 64) */
int cst_im2col_b(const float* e, void* col_bf16, int B, int L, int E, int k, int mode, int R, void* stream);
int cst_col2im(const float* dcol, float* de, int B, int L, int E, int k, int mode, int R, int accumulate, void* stream);

/* One filter size of the RelGAN_D convolution bank, fused: Conv2d(1, F, (k, E/R), stride (1, E/R)) over
 * the R representations of e [B, L, E], relu, max over time (discriminator.py:21-24, 41-44).
 *   feats[g, f] = max_t relu(bias[f] + sum_{x<k, c<E/R} e[b, t+x, rep*E/R + c] * w[f, x*E/R + c]),  g = b*R + rep
 *   arg[g, f]   = first t attaining the maximum (int32 [B*R, F])
 * feats may be a column slice of the concatenated feature matrix (leading dimension ldf).
 * bwd_input:  de (+)= d feats/d e (the relu gate is feats > 0); bwd_weight: dw [F, k*E/R], db [F]
 * (workspace: cst_relconv_bwd_weight_workspace_floats floats of scratch, contents irrelevant).
 * Limits: k*E/R <= 40, (E/R) % 4 == 0, L-k+1 <= 128, F <= 320 -- status 1 outside them. */
int cst_relconv_fwd(const float* e, int B, int L, int E, int R, int k, const float* w, const float* bias, int F,
                    float* feats, long ldf, int* arg, void* stream);
int cst_relconv_bwd_input(const float* dfeats, long ldd, const float* feats, long ldf, const int* arg,
                          const float* w, int B, int L, int E, int R, int k, int F,
                          float* de, int accumulate, void* stream);
long cst_relconv_bwd_weight_workspace_floats(int B, int R, int k, int E, int F);
int cst_relconv_bwd_weight(const float* dfeats, long ldd, const float* feats, long ldf, const int* arg,
                           const float* e, int B, int L, int E, int R, int k, int F,
                           float* dw, float* db, float* workspace, long workspace_floats, void* stream);

/* max over the middle axis of x [G,T,F] with argmax (classifier.py:32; discriminator.py:42; match.py:41). */
int cst_seqmax_fwd(const float* x, float* out, long ldo, int* arg, int G, int T, int F, void* stream);
int cst_seqmax_bwd(const float* dout, long ldd, const int* arg, const float* y, long ldy, int relu_gate,
                   float* dx, int G, int T, int F, void* stream);
/* the same gradient in bf16 only ([G T, F] dense): the operand of the bf16 dgrad / weight-gradient products of the convolution */
int cst_seqmax_bwd_b(const float* dout, long ldd, const int* arg, const float* y, long ldy, int relu_gate,
                     void* dx_bf16, int G, int T, int F, void* stream);

int cst_dropout(const float* x, long ldx, float* out, long ldo, int R, int C,
                float drop_p, uint32_t drop_seed, uint32_t drop_stream, const void* drop_seed_dev, void* stream);
int cst_axpby(const float* a, long lda, float alpha, const float* b, long ldb, float beta,
              float* out, long ldo, int R, int C, void* stream);
/* out = x * (*s_dev): chain rule through a scalar loss whose upstream gradient lives on the device. */
int cst_scale_dev(const float* x, const float* s_dev, float* out, long n, void* stream);
/* dx = y > 0 ? dy * pos_scale : slope * dy */
int cst_act_bwd(const float* dy, const float* y, float slope, float pos_scale, float* dx, long n, void* stream);
/* highway gate of discriminator.py:45-46. */
int cst_highway_fwd(const float* h, const float* pred, float* out, long n, void* stream);
int cst_highway_bwd(const float* dout, const float* h, const float* pred, float* dh, float* dpred, long n, void* stream);

/* loss[0] = weight * mean(...): kind 0 = MSE vs t (or the constant tconst when t is null)
 * (main_pretrain.py:72; main_optimize.py:107), kind 1 = BCE-with-logits vs tconst
 * (main_optimize.py:108,122-123); dx = gscale * d mean / dx. */
int cst_small_loss(const float* x, const float* t, float tconst, int kind, long n, float weight, float* loss,
                   float* dx, float gscale, void* stream);

/* Trainer(gradient_clip_val) + torch.optim.Adam (main_pretrain.py:61-64,139; main_warmup.py:41-43,103;
 * main_optimize.py:73-88,211), all on the device: out += sum(g^2); g *= min(1, max_norm/(sqrt(sumsq)+1e-6));
 * Adam with bias correction from the device step counter.  The sum of squares is deterministic (per-block partials in
 * `partials`, >= 1024 floats of caller scratch, added in index order by one block): data-parallel replicas must compute
 * bit-identical clip coefficients from bit-identical (all-reduced) gradients. */
int cst_sumsq_accumulate(const float* g, long n, float* out, float* partials, void* stream);
int cst_clip_scale(float* g, long n, const float* sumsq_dev, float max_norm, void* stream);
int cst_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                  const int* step_dev, void* stream);
/* cst_clip_scale + cst_adam_step in one pass for a group that steps right after the clip: every gradient is read as
 * g * min(1, max_norm / (sqrt(*sumsq_dev) + 1e-6)) -- bit for bit what cst_clip_scale would have stored -- and is left unscaled in
 * memory (the caller zeroes it after the step, main_pretrain.py / main_warmup.py / main_optimize.py optimizer steps). */
int cst_adam_step_clipped(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
                          const int* step_dev, const float* sumsq_dev, float max_norm, void* stream);
int cst_add_i32(int* p, int inc, void* stream);
/* flat[dst_off[t] + i] (+)= srcs[t][i] for all tensors in one launch (null source = skipped);
 * chunk tables (4096 elements per chunk) are built by the host once per parameter set.  ssq_partials (optional, [nchunks]): every chunk's
 * sum of squares of what its slot holds AFTER the call (skipped tensors included), so that the clip_grad_norm_ that follows
 * (main_pretrain.py:139, main_warmup.py:103, main_optimize.py:211) needs no pass of its own: cst_sumsq_partials adds them in index order. */
int cst_multi_accumulate(const void* srcs_dev, const long* dst_off_dev, const long* sizes_dev,
                         const int* chunk_tensor_dev, const long* chunk_start_dev, int nchunks,
                         float* flat, int accumulate, float* ssq_partials, void* stream);
int cst_sumsq_partials(const float* partials, int n, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CST_HIP_H */
