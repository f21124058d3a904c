/* libbmp_hip -- C ABI of the MI355X-native GCN-BMP hot path (gfx950 only).
 *
 * The reference (Minys233/GCN-BMP) has no FFI: its operator boundary is the Python call
 * protocol of chainer.Chain subclasses.  These entry points are what a
 * binding for that path would call -- one per reference operator -- and are what
 * gcn-bmp_amd/bmp/_lib.py binds with ctypes.  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc / torch caching allocator), 16-byte aligned;
 *     float = IEEE fp32, int = int32.  The library never allocates, frees, retains a pointer
 *     past the call, synchronises the host or keeps result-bearing global state: a call only
 *     enqueues kernels on `stream`.  What the library does remember, process-wide: (i) which
 *     (kernel, device) pairs already had hipFuncAttributeMaxDynamicSharedMemorySize raised
 *     (lock-free, per device: one process may drive several devices, from several host threads);
 *     (ii) A/B switches read once from the environment (BMP_WGRAD_DMA, BMP_WGRAD_XCD,
 *     BMP_STEP_WGRAD_UNFUSED, BMP_ROWGEMM_FORM / _DIRECT: diagnostics of tools/ and tests/,
 *     BMP_ROWGEMM_NO_THIN, BMP_ROWGEMM_SCALAR_EPI);
 *     (iii) the opt-in event timer bmp_prof_*.  None of them changes a result.  The code object
 *     also holds one read-only device array of 128 zero floats (what a listed weight-gradient
 *     problem reads past the end of its list).
 *   - row-indexed arrays use the packed layout of bmp/packed.py: N = n_tiles * bmp_tile_rows()
 *     rows (128 per tile, molecules never straddle a tile, dead rows are zero-weight).
 *   - weights come in two layouts: "T" = K-major [in x out] (used as the GEMM B operand) and
 *     "nat" = the reference Linear layout [out x in] (used by the backward-data GEMM).
 *   - return value 0 = ok; > 0 = hipError_t of a failed launch; < -1000 = argument check failed
 *     at source line -(ret + 1000).
 *   - `ws` is caller-provided scratch of at least *_ws_floats() floats.
 */
#ifndef BMP_H_
#define BMP_H_
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* bmp_stream_t; /* == hipStream_t */

enum { BMP_ACT_NONE = 0, BMP_ACT_SIGMOID = 1, BMP_ACT_TANH = 2, BMP_ACT_RELU = 3 };

int bmp_version(void);
int bmp_tile_rows(void);

/* Diagnostic (bench.py roofline leg; no reference counterpart): bracket every launch of one
 * kernel class with HIP events on its launch stream.  kclass: 1 row GEMM, 2 weight-gradient
 * GEMM, 3 gather, 4 co-attention pair kernel, 5/6 fused GGNN step fwd/bwd.  bmp_prof_stop waits
 * for the launches, returns their count and fills out[0..2] = total ms, executed flops, bytes. */
int bmp_prof_start(int kclass);
int bmp_prof_stop(double* out);
/* kclass -1 arms every class; bmp_prof_collect then reports per kernel: key = class * 16 + kernel id (csrc/bmp_kernels.h),
 * launches, total ms, executed flops and bytes as the launchers state them, into arrays of `cap` entries; returns the
 * number of distinct keys. */
int bmp_prof_collect(int* key, int* count, double* ms, double* flops, double* bytes, int cap);

/* A stream of the device's lowest priority for launches that run beside a dependent chain (no reference counterpart:
 * chainer runs one stream).  The backward's weight-gradient GEMMs go there: nothing in the chain reads their output, and
 * their workgroups fill the CUs the tile kernels' last round leaves idle.  Destroy with bmp_stream_destroy. */
/* bmp_msg_bwd, bmp_gru_bwd, bmp_readout_bwd, bmp_coattn_nie_bwd and bmp_mlp_bwd take that stream as `stream_w` (NULL or == stream: everything in line):
 * their weight-gradient launches go there, ordered behind what `stream` has been given up to the point inside the call
 * where their operands are complete.  The workspace `ws` is then read on both streams: the caller keeps it (and the row
 * tensors) alive until the two streams have joined.  accumulate_w != 0: the weight gradients add into their outputs (the
 * later calls of a tied layer).  bmp_coattn_nie_bwd launches its pair kernels once per size class of the drug pairs; a
 * class of at most 64 pairs next to a more populated one also runs on `stream_w`, beside the other classes, and `stream`
 * picks up behind it inside the call. */
int bmp_stream_create_low(bmp_stream_t* out);
int bmp_stream_destroy(bmp_stream_t stream);

/* EmbedAtomID lookup -- chainer_chemistry EmbedAtomID used at models/ggnn.py:85,603
 * (models/relgcn.py:40,67): out[row,:] = W[ids[row],:], W [V x d].  Ids outside [0, V) are the caller's error (the Python
 * wrappers raise ValueError before any launch, as chainer's EmbedID type check does); the kernels never leave the table:
 * such a row reads W[0] and contributes nothing to dW. */
int bmp_embed_fwd(const int* ids, const float* W, int N, int d, int V, float* out, bmp_stream_t stream);
size_t bmp_embed_bwd_ws_floats(int N, int d, int V);
int bmp_embed_bwd(const int* ids, const float* dout, int N, int d, int V, float* dW, float* ws, size_t ws_floats,
                  bmp_stream_t stream);   /* dW is overwritten; two deterministic passes through ws */

/* Message function -- GGNN.update message part models/ggnn.py:215-243 (= GGNNUpdate
 * models/update/ggnn_update.py:31-50); with WsT/bs/act=TANH the whole RelGCN layer
 * models/update/relgcn_update.py:24-44 + models/relgcn.py:71.
 *   out = act( gather_sum(x) . WT + wdeg . bE [+ x . WsT + bs] )
 * WT [4*d_in x d_out] row e*d_in+k, col c == reference W[4c+e][k]; bE [4 x d_out] == b[4c+e].
 * Saves agg [N x 4*d_in] and wdeg [N x 4] for the backward. */
int bmp_msg_fwd(const float* x, int ldx, int n_tiles, int d_in, int d_out, const int* csr_ptr, const int* csr_col,
                const float* csr_val, const float* WT, const float* bE, const float* WsT, const float* bs, int act,
                float* agg, float* wdeg, float* out, int ldo, bmp_stream_t stream);
size_t bmp_msg_bwd_ws_floats(int n_tiles, int d_in, int d_out);
int bmp_msg_bwd(const float* dout, int lddo, const float* out, int ldo, int act, const float* x, int ldx, int n_tiles,
                int d_in, int d_out, const int* csrT_ptr, const int* csrT_col, const float* csrT_val, const float* Wnat,
                const float* Ws, const float* agg, const float* wdeg, float* dx, float* dWT, float* dbE, float* dWsT,
                float* dbs, int accumulate_w, const int* type_rows_f, const int* type_cnt_f, float* ws, size_t ws_floats,
                bmp_stream_t stream, bmp_stream_t stream_w);
/* (type_rows_f / type_cnt_f: optional bmp_type_rows lists of the FORWARD CSR -- the rows whose gathered features agg_e are not
 *  zero: dWT = agg^T . dpre then sums over them only, per bond type) */

/* GRU node update -- chainer links.GRU (StatefulGRU) at models/ggnn.py:132,254-262
 * (models/update/ggnn_update.py:28,61).  first != 0 selects the first-call-after-reset branch.
 * AT [2d x 3d] rows [h;m] cols [r|z|c] (state terms folded by the caller), UcT [d x d], b [3d].
 * Saves rz [N x 2d], c [N x d]. */
int bmp_gru_fwd(const float* h, const float* m, int n_tiles, int d, int first, const float* AT, const float* UcT,
                const float* b, float* rz, float* c, float* hout, bmp_stream_t stream);
size_t bmp_gru_bwd_ws_floats(int n_tiles, int d);
int bmp_gru_bwd(const float* dhout, const float* h, const float* m, const float* rz, const float* c, int n_tiles, int d,
                int first, const float* A, const float* Uc, float* dh, float* dm, float* dAT, float* dUcT, float* db,
                int accumulate_w, float* ws, size_t ws_floats, bmp_stream_t stream, bmp_stream_t stream_w);

/* The same update with the GRU state apart from its input: dropout on the step output (models/ggnn.py:626-627) feeds the
 * next step x = [hd, m] with hd = dropout(s) while the stateful GRU keeps the un-dropped s.  Later calls only (the first
 * call after reset has no state: bmp_gru_fwd with first = 1).  WT [2d x 3d] = [W_r | W_z | W]^T (rows [hd-part; m-part]),
 * UrzT [d x 2d] = [U_r | U_z]^T, UcT [d x d] = U^T, b [3d] = bW + bU; A, Urz, Uc: their transposes (reference layouts). */
int bmp_gru_state_fwd(const float* hd, const float* m, const float* s, int n_tiles, int d, const float* WT, const float* UrzT,
                      const float* UcT, const float* b, float* rz, float* c, float* sout, bmp_stream_t stream);
size_t bmp_gru_state_bwd_ws_floats(int n_tiles, int d);
int bmp_gru_state_bwd(const float* dsout, const float* hd, const float* m, const float* s, const float* rz, const float* c,
                      int n_tiles, int d, const float* A, const float* Urz, const float* Uc, float* dhd, float* dm, float* ds,
                      float* dWT, float* dUrzT, float* dUcT, float* db, float* ws, size_t ws_floats, bmp_stream_t stream);

/* One whole GGNN propagation step, fused per 128-row tile -- GGNN.update models/ggnn.py:215-263
 * (message + GRU in one kernel, the tile's atom states resident in LDS).  d must satisfy
 * bmp_ggnn_step_supported(d) (64 or 128); other widths use bmp_msg_* + bmp_gru_*.
 * Weights have the shapes of bmp_msg_fwd / bmp_gru_fwd but are "K4-packed": a K-major [K x N] matrix
 * is stored as [K/4][N][4] (element (k, n) at ((k/4)*N + n)*4 + k%4), so one lane's four consecutive
 * k values of a column are a single 16-byte load (WT, AT, UcT for fwd; Wnat [d x 4d], A [3d x 2d],
 * Uc [d x d] for bwd).  fwd saves m [N x d], rz [N x 2d], c [N x d].
 * bwd writes dh [N x d] and gda [N x 7d] = [G_0..G_3 (transposed-gathered dm per bond type) | da_r | da_z | da_c];
 * wgrad reduces over the N rows: o1 [d x 7d] = h^T.gda (cols [0,4d): dWT as [k][e*d+c]; cols [4d,7d): dAT rows
 * 0..d-1), o2 [d x 3d] = m^T.da (dAT rows d..2d-1), dUcT [d x d], cs [7d] = column sums (dbE | db); all of it is ONE
 * GEMM launch + ONE fixed-order reduction.  first != 0 (the GRU's first call after reset has no r gate): the da_r columns
 * of gda are neither written by bwd nor read by wgrad and count as zeros. */
/* The forward launches take a tile range: tiles tile0 .. tile0 + n_tiles - 1 of the WHOLE arrays passed (a step is
 * tile-local: molecules never straddle tiles), so that two halves of a batch can run as two chains on two streams.
 * mt_row0 / mt_nblk (both NULL: tile t = rows [128 t, 128 t + 128)): the tile table of the encoder layout
 * (bmp_collate_plan_enc) -- tile t = rows [mt_row0[t], mt_row0[t] + 32 mt_nblk[t]), 1 <= mt_nblk <= 4; a tile's dead
 * blocks cost no gather, MFMA, load or store; mt_rows = the rows of the launch's tiles (the event timer's flop accounting,
 * ignored without a table).  m, rz, c may be NULL together (forward-only evaluation).
 * tile_stride (bmp_ggnn_step_fwd / _bwd; 0 or 128, 128 only with a table): 0 = dense rows, tile t + 1 follows tile t;
 * 128 = the tiles sit at a fixed stride (a batch of a fixed shape whose step is recorded once as a HIP graph: one molecule
 * per tile, mt_nblk its live blocks) and every launch CLEARS the rows of a tile's other blocks in the arrays it writes
 * (hout, m, rz, c; dh, gda), so that nothing an earlier batch left there enters the GEMMs that walk all rows. */
int bmp_ggnn_step_supported(int d);
int bmp_ggnn_step_fwd(const float* h, int tile0, int n_tiles, int d, int first, const int* csr_ptr, const int* csr_col,
                      const float* csr_val, const float* WT, const float* bE, const float* AT, const float* UcT,
                      const float* b, float* m, float* rz, float* c, float* hout, const int* mt_row0, const int* mt_nblk,
                      int mt_rows, int tile_stride, bmp_stream_t stream);
int bmp_ggnn_step_bwd(const float* dhout, const float* h, const float* rz, const float* c, int n_tiles, int d, int first,
                      const int* csrT_ptr, const int* csrT_col, const float* csrT_val, const float* Wnat, const float* A,
                      const float* Uc, float* dh, float* gda, const int* mt_row0, const int* mt_nblk, int mt_rows, int tile_stride,
                      int skip_zero_g, bmp_stream_t stream);
/* skip_zero_g != 0 (bmp_ggnn_step_bwd, bmp_relgcn_layer_bwd): the caller reads gda's per-type blocks through the batch's row lists
 * only -- it passes type_rows to the wgrad call and bmp_step_wgrad_lists_used(N, d) said the lists will be used -- so the block
 * of a row without a bond of the type, an exact zero, is not written. */
int bmp_step_wgrad_lists_used(int N, int d);
size_t bmp_ggnn_step_wgrad_ws_floats(int N, int d);
/* type_rows [4 x N] / type_cnt [4] (both optional, NULL together): bmp_type_rows of the batch's TRANSPOSED CSR.  A row's
 * gathered gradient G_e is an exact zero unless the row has a bond of type e (73 / 19 / 2 / 52 % of the rows of a DDI batch for
 * single / double / triple / aromatic): with the lists the four per-type blocks of o1 sum over their rows only.
 * live_rows / live_cnt (optional, NULL together; used with type_rows): the fifth list of bmp_type_rows_live, the rows that
 * belong to a molecule -- for a batch most of whose rows belong to none (tiles at a fixed stride) the gate blocks of o1 and o2
 * sum over that list instead of all N rows (later calls only; the other rows' gda is zero either way). */
int bmp_ggnn_step_wgrad(const float* h, const float* m, const float* rz, const float* gda, int N, int d, int first,
                        float* o1, float* o2, float* dUcT, float* cs, int accumulate, const int* type_rows, const int* type_cnt,
                        const int* live_rows, const int* live_cnt, float* ws, size_t ws_floats, bmp_stream_t stream);

/* Rows by bond type of a CSR (no reference counterpart: the reference's dense (mb, 4, A, A) adjacency multiplies the zeros,
 * models/ggnn.py:229-242): idx[e * N + p] = the p-th row, ascending, that holds an entry of type e; cnt[e] = their number
 * (device arrays; the count never visits the host).  ws: bmp_type_rows_ws_ints(N) ints.  Fixed order, no atomics. */
size_t bmp_type_rows_ws_ints(int N);
int bmp_type_rows(const int* csr_ptr, const int* csr_col, int N, int* idx, int* cnt, int* ws, bmp_stream_t stream);
/* The same with a fifth list, idx [5 x N] / cnt [5]: list 4 = the rows that belong to a molecule (row_mol >= 0), ascending. */
int bmp_type_rows_live(const int* csr_ptr, const int* csr_col, const int* row_mol, int N, int* idx, int* cnt, int* ws,
                       bmp_stream_t stream);

/* One whole RelGCN layer, fused per 128-row tile -- RelGCNUpdate.__call__ models/update/relgcn_update.py:24-44 with
 * the tanh of models/relgcn.py:71: out = act(h W_s^T + b_s + sum_e adj'_e (W_e h + b_e)), the 1/degree of rescale_adj
 * (models/relgcn.py:20-28) carried by the CSR values.  d_in == d_out == d with bmp_relgcn_layer_supported (64 or 128);
 * other shapes use bmp_msg_fwd/bwd.  WT [4d x d], WsT [d x d] (fwd) and Wnat [d x 4d] = WT^T, Ws [d x d] = WsT^T (bwd)
 * are K4-packed as for bmp_ggnn_step_*.  fwd saves wdeg [N x 4] (weighted degree per bond type).
 * bwd writes dh [N x d] and gda [N x 5d] = [G_0..G_3 (transposed-gathered dpre per bond type) | dpre].
 * wgrad: o1 [d x 5d] = h^T.gda (cols [0,4d): dWT as [k][e*d+c]; cols [4d,5d): dWsT), dbE [4 x d] = wdeg^T.dpre,
 * cs [5d] = column sums of gda (cs[4d:] = dbs). */
int bmp_relgcn_layer_supported(int d_in, int d_out);
int bmp_relgcn_layer_fwd(const float* h, int tile0, int n_tiles, int d, const int* csr_ptr, const int* csr_col, const float* csr_val,
                         const float* WT, const float* bE, const float* WsT, const float* bs, int act, float* out,
                         float* wdeg, const int* mt_row0, const int* mt_nblk, int mt_rows, bmp_stream_t stream);   /* wdeg may be NULL */
int bmp_relgcn_layer_bwd(const float* dout, const float* out, int act, int n_tiles, int d, const int* csrT_ptr,
                         const int* csrT_col, const float* csrT_val, const float* Wnat, const float* Ws, float* dh,
                         float* gda, const int* mt_row0, const int* mt_nblk, int mt_rows, int skip_zero_g, bmp_stream_t stream);
size_t bmp_relgcn_layer_wgrad_ws_floats(int N, int d);
int bmp_relgcn_layer_wgrad(const float* h, const float* wdeg, const float* gda, int N, int d, float* o1, float* dbE,
                           float* cs, int accumulate, const int* type_rows, const int* type_cnt, float* ws, size_t ws_floats,
                           bmp_stream_t stream);          /* type_rows / type_cnt: as bmp_ggnn_step_wgrad */

/* Gated-sum readout -- GGNN.readout models/ggnn.py:333-341 and GGNNReadout.__call__
 * models/readout/ggnn_readout.py:42-57.  g[mol] = sum_rows w * sigmoid(i(.)) * act_j(j(.)).
 * WT [(d+d0) x 2o] cols [i|j]; h0 may be NULL (d0 ignored).  Saves ij [N x 2o]. */
int bmp_readout_fwd(const float* h, const float* h0, int n_tiles, int d, int d0, int o, const float* WT, const float* b,
                    int act_j, const float* row_w, const int* mol_row0, const int* mol_nrows, int n_mols, float* ij,
                    float* g, bmp_stream_t stream);
size_t bmp_readout_bwd_ws_floats(int n_tiles, int d, int d0, int o);
int bmp_readout_bwd(const float* dg, const float* h, const float* h0, int n_tiles, int d, int d0, int o,
                    const float* Wnat, const float* ij, int act_j, const float* row_w, const int* mol_row0,
                    const int* mol_nrows, int n_mols, float* dh, float* dh0, float* dWT, float* db, int accumulate_w,
                    float* ws, size_t ws_floats, bmp_stream_t stream, bmp_stream_t stream_w);
/* The same forward as one kernel per tile when bmp_readout_tile_supported(d, d0, o) (d == o in {64, 128}, d0 in {0, d}):
 * WT K4-packed as for bmp_ggnn_step_*, row_mol [N] = molecule of every packed row (-1: none).  Every molecule lies in
 * one tile, whose workgroup takes its sum in a fixed order. */
int bmp_readout_tile_supported(int d, int d0, int o);
int bmp_readout_tile_fwd(const float* h, const float* h0, int n_tiles, int d, const float* WT, const float* b, int act_j,
                         const float* row_w, const int* row_mol, const int* mol_nrows, float* ij, float* g,
                         bmp_stream_t stream);

/* GraphLinear on row tiles -- chainer_chemistry GraphLinear (models/ggnn.py:16,88,135,139;
 * nie_coattention.py:325-329): Y = act(X . WT + b); weight/bias gradients over N rows. */
int bmp_linear_fwd(const float* X, int ldx, int n_tiles, int K, int Nout, const float* WT, int ldw, const float* b,
                   int act, float* Y, int ldy, bmp_stream_t stream);
size_t bmp_wgrad_ws_floats_c(int N, int K, int Nn);
int bmp_linear_wgrad(const float* X, int ldx, const float* dY, int ldy, int N, int K, int Nn, float* dWT, float* db,
                     float* ws, size_t ws_floats, bmp_stream_t stream);

/* Per-molecule segment operators -- the atom x molecule-vector arithmetic of the coarse co-attention family:
 * ParallelCoattention parallel_coattention.py:34-84, AlternatingCoattention alternating_coattention.py:36-86,
 * GlobalCoattention global_coattention.py:27-73, NeuralCoattention neural_coattention.py:27-71.
 *   segpool   : out[m,c] = sum_rows w * A[r, c|0] * Y[r,c]            (ca = 1: per-atom scalar weight; ca = o: gate)
 *   segsoftmax: alpha[r] = softmax over the molecule's atoms of s[r] (multiplicities in the denominator)
 *   rowbcast  : out[r,:] = q[row_mol[r],:]  (tile the other molecule's vector over the atoms); bwd = per-molecule sum
 *   rowdot    : s[r] = x[r,:] . u[row_mol[r],:] + s0[row_mol[r]]      (Bilinear / matmul of atom and molecule vector) */
int bmp_segpool_fwd(const float* A, int ca, const float* Y, int o, const float* w, const int* mol_row0,
                    const int* mol_nrows, int n_mols, float* out, bmp_stream_t stream);
int bmp_segpool_bwd(const float* dout, const float* A, int ca, const float* Y, int o, const float* w, const int* mol_row0,
                    const int* mol_nrows, int n_mols, int N, float* dA, float* dY, bmp_stream_t stream);
int bmp_segsoftmax_fwd(const float* s, const float* w, const int* mol_row0, const int* mol_nrows, int n_mols, int N,
                       float* alpha, bmp_stream_t stream);
int bmp_segsoftmax_bwd(const float* dalpha, const float* alpha, const float* w, const int* mol_row0, const int* mol_nrows,
                       int n_mols, int N, float* ds, bmp_stream_t stream);
int bmp_rowbcast_fwd(const float* q, int c, const int* row_mol, int N, float* out, bmp_stream_t stream);
int bmp_rowbcast_bwd(const float* d, int c, const int* mol_row0, const int* mol_nrows, int n_mols, float* dq,
                     bmp_stream_t stream);
int bmp_rowdot_fwd(const float* x, int d, const float* u, const float* s0, const int* row_mol, int N, float* s,
                   bmp_stream_t stream);
int bmp_rowdot_bwd(const float* ds, const float* x, int d, const float* u, const int* row_mol, const int* mol_row0,
                   const int* mol_nrows, int n_mols, int N, float* dx, float* du, float* ds0, bmp_stream_t stream);
/* CircularParallelCoattention.circular_correlation (models/coattention/parallel_coattention.py:162-187) of every atom
 * row with its molecule's vector: e[r][k] = sum_t a[r][t] * q[mol(r)][(t + k) mod o].  a, e, de, da [N x o] packed
 * rows (rows of no molecule are not written), q, dq [n_mols x o]. */
int bmp_rowcorr_fwd(const float* a, int o, const float* q, const int* mol_row0, const int* mol_nrows, int n_mols, float* e,
                    bmp_stream_t stream);
int bmp_rowcorr_bwd(const float* de, const float* a, int o, const float* q, const int* mol_row0, const int* mol_nrows,
                    int n_mols, float* da, float* dq, bmp_stream_t stream);

/* Fine-grained co-attention -- NieFineCoattention.__call__ + compute_attention
 * models/coattention/nie_coattention.py:335-396 (VQAParallelCoattention
 * vqa_parallel_coattention.py:42-102 is the same computation).  Pair b couples rows
 * [r1[b], r1[b]+n1[b]) of X1 with rows [r2[b], r2[b]+n2[b]) of X2; w1/w2 are the row multiplicities.
 * WbT[q*d+p] = W[p][q] (bilinear form), ZW{1,2}T [d x ZC] = [Wj^T | Wl_k^T | V_k | 0] with
 * ZC = bmp_coattn_zcols(o, H), zb [ZC] = [bj | 0], wa{1,2} [H], cbias [1].
 * `order` lists the pair ids grouped by size class (max(n1,n2) <= 32, 64, 96, 128, and `nbig` pairs with a molecule of
 * more than 128 rows: the reference's preprocessor has no size limit, train_ddi_modify.py:256) with n32..n128, nbig the
 * class counts (sum = B): one launch per class, LDS sized by the class -- the forward takes the classes in DESCENDING size
 * (oversized pairs first), the backward ascending.  The oversized class runs the same program out of a global workspace
 * (bmp_coattn_big_ws_floats; np_big = the largest row count among its pairs; ws_big may be NULL when nbig == 0).
 * Saves Q2 [N2 x d], Z1/Z2 [N x ZC], H1/H2 [N x H], al1/al2 [N] and Cbuf: pair b owns n2*n1 + 2*(n1 + n2) floats at
 * coff[b] -- C (n2 x n1, row-major), then the softmax statistics of C the backward reloads: cmax [n1], 1/D2 [n1]
 * (column softmax over side-2 atoms), rmax [n2], 1/D1 [n2] (row softmax over side-1 atoms). */
int bmp_coattn_zcols(int o, int H);
int bmp_coattn_nie_fwd(const float* X1, int n_tiles1, const float* X2, int n_tiles2, int d, int o, int H, int act, int mode,
                       const float* w1, const int* r1, const int* n1, const float* w2, const int* r2, const int* n2,
                       const long long* coff, int B, const int* order, int n32, int n64, int n96, int n128, int nbig, int np_big,
                       const float* WbT, const float* ZW1T, const float* ZW2T,
                       const float* zb, const float* wa1, const float* wa2, const float* cbias, float* Q2, float* Z1,
                       float* Z2, float* Cbuf, float* H1, float* H2, float* al1, float* al2, float* out1, float* out2,
                       float* ws_big, size_t ws_big_floats, bmp_stream_t stream);
size_t bmp_coattn_big_ws_floats(int np_big, int H, int o, int nbig, int backward);
size_t bmp_coattn_nie_bwd_ws_floats(int n_tiles1, int n_tiles2, int d, int o, int H, int B, int nbig, int np_big);
int bmp_coattn_nie_bwd(const float* dout1, const float* dout2, const float* X1, int n_tiles1, const float* X2,
                       int n_tiles2, int d, int o, int H, int act, int mode, const float* w1, const int* r1, const int* n1,
                       const float* w2, const int* r2, const int* n2, const long long* coff, int B, const int* order,
                       int n32, int n64, int n96, int n128, int nbig, int np_big, const float* Wb, const float* ZW1, const float* ZW2,
                       const float* wa1, const float* wa2,
                       const float* Q2, const float* Z1, const float* Z2, const float* Cbuf, const float* H1,
                       const float* H2, const float* al1, const float* al2, float* dX1, float* dX2, float* dWbT,
                       float* dZW1T, float* dZW2T, float* dzb, float* dwa, float* ws, size_t ws_floats,
                       bmp_stream_t stream, bmp_stream_t stream_w, const int* row_mol1, const int* row_mol2,
                       const float* gscale);

/* ---- BiMPM matching -- models/coattention/bimpm.py:45-199 with aggr = F.sum (train_binary.py:253-256) ----
 * mol_1, mol_2 [B x 3H] for B drug pairs: max-pooling matching, attentive-mean matching and attentive-max matching of every
 * atom against the other molecule, `H` perspectives each, summed over the atoms (with the row multiplicities w of the packed
 * layout; maxima over rows with w > 0).  X1 / X2 [N x d]: packed atom rows of the two sides; r / n: first row and row count
 * of every pair's molecule (n <= maxn); P, Q, R [H x d] = max_pooling_W, att_mean_W, att_max_W.  The backward recomputes the
 * forward (same order: the same maxima win) and overwrites dP, dQ, dR and the rows of dX1 / dX2 that belong to a pair. */
int bmp_bimpm_supported(int d, int H, int maxn);
size_t bmp_bimpm_ws_floats(int d, int H, int maxn, int B, int backward);
int bmp_bimpm_fwd(const float* X1, const float* X2, int d, int H, const float* w1, const int* r1, const int* n1, const float* w2,
                  const int* r2, const int* n2, int B, int maxn, const float* P, const float* Q, const float* R, float* out1,
                  float* out2, float* ws, size_t ws_floats, bmp_stream_t stream);
int bmp_bimpm_bwd(const float* dout1, const float* dout2, const float* X1, const float* X2, int d, int H, const float* w1,
                  const int* r1, const int* n1, const float* w2, const int* r2, const int* n2, int B, int maxn, const float* P,
                  const float* Q, const float* R, float* dX1, float* dX2, float* dP, float* dQ, float* dR, float* ws,
                  size_t ws_floats, bmp_stream_t stream);

/* ---- the reference's dense batch on the device (concat_mols, train_ddi_modify.py:296) ----
 * adj (mb, 4, A, A) float32, zero padded.  bmp_dense_count: bonds per dense position, by row (incoming) and by column
 * (outgoing).  bmp_dense_to_csr: the entries of the packed CSR of bmp/packed.py (transposed = 1: of its transpose), in
 * the host packer's order; rowmap [mb x A] = packed row of every dense position, ptr [N + 1] = row pointers. */
int bmp_dense_count(const float* adj, int mb, int A, int* row_nnz, int* col_nnz, bmp_stream_t stream);
int bmp_dense_to_csr(const float* adj, int mb, int A, const int* rowmap, const int* ptr, int transposed, int* col, float* val,
                     bmp_stream_t stream);

/* ---- the per-iteration collate for a drug store resident in HBM (SerialIterator + converter=concat_mols,
 * train_ddi_modify.py:280,295-296; the batch contract of SURVEY.md 8(a) R0 from index pairs) ----
 * bmp_collate_plan and bmp_collate_pair_meta are HOST functions (host pointers, no device work, no stream): the size
 * arithmetic of concat_mols (zero-padding width per batch side) and of the packed layout (first-fit-decreasing tile
 * placement, entry bases, dead rows), and the co-attention's per-pair metadata.  st_nrows / st_nedges [n_store]: rows
 * (real atoms + 1) and directed bonds of every molecule of the store; mids: molecule of every instance, the sides of the
 * batch one after the other (side_ptr [n_sides + 1]); pad_to [n_sides] or NULL.  tab (out, 6 * I int32):
 * row0 | nrows | mid | ebase | padw | ndead.  side_tiles (out) [n_sides + 1]; totals (out) [4]: n_tiles, n_edges,
 * n_real_atoms, max rows of an instance.  meta (out, 8 * B int32, 8-byte aligned): coff (int64) | r1 | n1 | r2 | n2 |
 * order | order_f as bmp_coattn_nie_fwd/_bwd take them; counts [6]: the five size classes and np_big; ctotal.
 * An instance of more than R rows takes ceil(rows / R) whole consecutive tiles at the head of its side.
 * bmp_collate_emit (DEVICE): writes the packed batch of bmp/packed.py from the plan table and the store's arrays
 * (st_rowoff / st_eoff [M + 1], st_atom, per-row local entry ends st_rend / st_rendT, local entries st_col / st_colT =
 * local row << 2 | bond type).  Bit-identical to the host packer. */
int bmp_collate_plan(const int* st_nrows, const int* st_nedges, int n_store, const int* mids, const int* side_ptr,
                     int n_sides, const int* pad_to, int R, int* tab, int* side_tiles, long long* totals);
int bmp_collate_pair_meta(const int* tab, int I, int B, int side1_tiles, int R, int* meta, int* counts, long long* ctotal);
int bmp_collate_emit(const int* tab, int I, const int* st_rowoff, const int* st_eoff, const int* st_atom, const int* st_rend,
                     const int* st_rendT, const int* st_col, const int* st_colT, int* atom_id, float* row_w, int* row_mol,
                     int* csr_ptr, int* csr_col, float* csr_val, int* csrT_ptr, int* csrT_col, float* csrT_val,
                     const int* tile_last, bmp_stream_t stream);   /* tile_last: NULL, or bmp_collate_plan_enc's (pad-row marks) */

/* ---- the encoder's own row layout (csrc/bmp_enc.hip, bmp/enclayout.py) ----
 * Inside the encoder the zero-padded positions of a batch (concat_mols, train_ddi_modify.py:296; unmasked everywhere,
 * models/ggnn.py:340,603) all carry ONE state per propagation step: atom id 0, no bonds (models/ggnn.py:215-263).  The encoder
 * layout holds the real atoms of every encoded molecule and one pad row per tile, in tiles of 1..4 live 32-row blocks sized so
 * that the CUs finish together (bmp_collate_plan_enc, a HOST function like bmp_collate_plan); with dedup != 0 a molecule that
 * occurs several times in the batch is encoded once (SURVEY.md 8(d) caveat; at most 544 distinct drugs, setting.py:30).  The
 * readout and the co-attention keep the per-instance layout; two index kernels connect the layouts:
 *   bmp_encrows_expand: out[instance row] = h[encoder row of that atom] (an instance's pad row <- its tile's pad row; rows of
 *                       no instance: 0);
 *   bmp_encrows_reduce: dh[encoder row] = sum of dX over the instance rows copied from it, in a fixed order (no atomics).
 * Table meanings: see csrc/bmp_collate.hip (bmp_collate_plan_enc) and csrc/bmp_enc.hip.  bmp_collate_plan_enc returns -2000
 * when a molecule has more than 127 atoms (the batch then keeps the per-instance layout). */
int bmp_collate_plan_enc(const int* st_nrows, const int* st_nedges, int n_store, const int* mids, int I, int dedup, int n_cu,
                         int R, int* tab, int* tile_last, int* uid, int* uptr, int* uinst, int* enc_pad, int* tptr, int* tmols,
                         int* mt_row0, int* mt_nblk, long long* totals);
int bmp_encrows_expand(const float* h, int d, const int* row_mol, const int* inst_row0, const int* uid, const int* enc_row0,
                       const int* enc_n, const int* enc_pad, int N_inst, float* out, bmp_stream_t stream);
int bmp_encrows_reduce(const float* dX, int d, const int* erow_mol, const int* enc_row0, const int* enc_n, const int* uptr,
                       const int* uinst, const int* inst_row0, const int* tptr, const int* tmols, int N_enc, float* dh,
                       bmp_stream_t stream);

/* rescale_adj -- models/relgcn.py:20-28 on the packed CSR: csr_val_out[e] = csr_val[e] * (1 / deg(source of e)), and the
 * same for the transposed CSR; deg = sum of the source atom's bond values over types and destinations (0 -> 1). */
int bmp_rescale_adj(const int* csr_col, const float* csr_val, int E, const int* csrT_ptr, const float* csrT_val, int N,
                    float* csr_val_out, float* csrT_val_out, bmp_stream_t stream);

/* ---- link predictor tail: MLP (models/mlp.py:20-45: Linear -> relu -> ... -> Linear on [g1 | g2], train_binary.py:98-101)
 * and sigmoid cross entropy (chainer.functions.sigmoid_cross_entropy, train_ddi_modify.py:285) ----
 * x = [x1 (B x d1) | x2 (B x d2)] (x2 NULL when d2 = 0); dims[0..nl] layer widths (dims[0] = d1 + d2 <= 1024, the others
 * <= 64, nl <= 4); W, b, act, dW, db are HOST arrays of nl device pointers: W[l] [dims[l+1] x dims[l]] (reference layout),
 * act[l] [B x dims[l+1]] (relu outputs, the last one = logits).  sce: loss[0] = mean over t != -1 of softplus(y) - t*y,
 * sums[2] = numerator, count (kept for the backward); dy = gout[0] * (sigmoid(y) - t) / count. */
int bmp_mlp_fwd(const float* x1, int d1, const float* x2, int d2, int B, int nl, const int* dims, const float* const* W,
                const float* const* b, float* const* act, bmp_stream_t stream);
size_t bmp_mlp_bwd_ws_floats(int B, int nl, const int* dims);
int bmp_mlp_bwd(const float* dy, const float* x1, int d1, const float* x2, int d2, int B, int nl, const int* dims,
                const float* const* W, float* const* act, float* dx1, float* dx2, float* const* dW, float* const* db,
                float* ws, size_t ws_floats, bmp_stream_t stream, bmp_stream_t stream_w);
/* The head of a TRAINING step in one launch -- what chainer_chemistry's Classifier(predictor, lossfun=F.sigmoid_cross_entropy)
 * does around the link predictor (train_ddi_modify.py:284-286; models/mlp.py:40-45): MLP forward (act as bmp_mlp_fwd, the last
 * entry = the logits), mean sigmoid cross entropy over the labels t [B x dims[nl]] != -1 (loss [1]; sums [2] = numerator |
 * count), its gradient dy [B x dims[nl]] and the MLP backward down to the input rows: dx1 / dx2 = d loss / d x for d loss = 1.
 * part: bmp_mlp_sce_ws_floats(B) floats; ticket: one unsigned, zero before the call and zero again after the launch.
 * bmp_mlp_bwd_w: the weight / bias gradients from that dy (bmp_mlp_bwd's, without dx), multiplied with the device scalar
 * gscale[0] (NULL: 1) -- the gradient that arrives at the loss; ws as bmp_mlp_bwd. */
size_t bmp_mlp_sce_ws_floats(int B);
int bmp_mlp_sce_fwdbwd(const float* x1, int d1, const float* x2, int d2, int B, int nl, const int* dims, const float* const* W,
                       const float* const* b, float* const* act, const int* t, float* dy, float* dx1, float* dx2, float* loss,
                       float* sums, float* part, unsigned* ticket, bmp_stream_t stream);
int bmp_mlp_bwd_w(const float* dy, const float* x1, int d1, const float* x2, int d2, int B, int nl, const int* dims,
                  const float* const* W, float* const* act, float* const* dW, float* const* db, const float* gscale, float* ws,
                  size_t ws_floats, bmp_stream_t stream);
int bmp_sce_fwd(const float* y, const int* t, int n, float* loss, float* sums, bmp_stream_t stream);
int bmp_sce_bwd(const float* y, const int* t, int n, const float* sums, const float* gout, float* dy, bmp_stream_t stream);

/* ---- pair features of the other link predictors (models/mlp.py:48-193, train_binary.py:102-116); the relu-MLP tail is
 * bmp_mlp_*.  kind: 0 SymMLP [g1+g2 | g1*g2] (:104-110), 1 HolE circular correlation c[k] = sum_i g1[i] g2[(i+k)%d]
 * (:126-151), 2 DistMult y[o] = sum_p W[o,p] g1[p] g2[p] (BilinearDiag :153-193, W [K x d]), 3 NTN = links.Bilinear
 * (:52,66: W [d1 x d2 x K], optional V1 [d1 x K], V2 [d2 x K], b [K]).  out / dout [B x bmp_pairfeat_cols]. */
int bmp_pairfeat_cols(int kind, int d, int K);
int bmp_pairfeat_fwd(int kind, const float* x1, const float* x2, int B, int d1, int d2, const float* W, const float* V1,
                     const float* V2, const float* b, int K, float* out, bmp_stream_t stream);
int bmp_pairfeat_bwd(int kind, const float* dout, const float* x1, const float* x2, int B, int d1, int d2, const float* W,
                     const float* V1, const float* V2, int K, float* dx1, float* dx2, float* dW, float* dV1, float* dV2,
                     float* db, bmp_stream_t stream);

/* ---- host glue of a training step (no counterpart kernels in the reference: there the layout changes are Chainer
 * function nodes and the optimizer is chainer.optimizers.Adam, train_ddi_modify.py:289) ----
 * bmp_gather_sum: dst[i] (=|+=) sum_k src[idx[k*n + i]] over table entries >= 0 (idx is [K][n] int32).  One launch
 * builds every kernel-layout weight array from the flat parameter buffer; one more folds the weight-gradient
 * buffers back into the flat gradient (bmp/plan.py builds the tables).
 * bmp_adam_step: m += (1-b1)(g-m); v += (1-b2)(g^2-v); p = p(1-wd) - alpha_t m/(sqrt(v)+eps), g scaled by grad_scale;
 * alpha_t_dev (device, may be NULL) overrides alpha_t: the step-dependent factor of a launch recorded in a HIP graph. */
int bmp_gather_sum(float* dst, int n, const float* src, const int* idx, int K, int accumulate, bmp_stream_t stream);
int bmp_adam_step(float* p, const float* g, float* m, float* v, int n, float alpha_t, const float* alpha_t_dev, float beta1,
                  float beta2, float eps, float weight_decay_rate, float grad_scale, bmp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* BMP_H_ */
