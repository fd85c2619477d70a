/*
 * jtvae_hip.h - C ABI of libjtvae_hip.so: the JT-VAE encoder message-passing
 * path (MPN, JTNNEncoder/GraphGRU, JTMPN) as hand-written HIP kernels for
 * MI355X (gfx950).
 *
 * The reference has no FFI: its encoder is Python calling stock ATen ops.  Each
 * entry point below replaces the group of ATen calls named in its comment
 * (file:line under the reference tree), so that a Python (ctypes), C or C++
 * host can drive the path with plain device pointers.
 *
 * Conventions
 *  - every pointer is DEVICE memory unless the parameter is documented "host";
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *    every call only enqueues work, none synchronises or allocates device memory
 *    (jtvae_tree_encoder_forward / _backward create a small ring of events per host
 *    thread and device the first time they fork their helper stream);
 *  - return value: 0 = ok, negative = argument rejected before any launch
 *    (JTVAE_E_*), positive = hipError_t from the launch;
 *  - "row buffers" hold one feature row per graph element, fp32, row-major with
 *    a leading dimension `ld` (floats).  The vectorised kernels need
 *    ld % 4 == 0 and a 16-byte aligned base; columns [H, ld) are padding, which
 *    every kernel here writes as zero and never lets leak into columns < H;
 *  - graph adjacency is CSR: rowptr[n_rows+1], col[nnz], int32, built once per
 *    batch from the reference's padded tables (agraph/bgraph/node_graph/
 *    mess_graph) with the padding zeros dropped.  Dropping them is exact because
 *    row 0 of every gathered buffer is the all-zero padding row in the
 *    reference's layout (fast_jtnn/mpn.py:112-114, fast_jtnn/jtnn_enc.py:108-109,
 *    fast_jtnn/jtmpn.py:83,95).
 */
#ifndef JTVAE_HIP_H
#define JTVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JTVAE_ABI_VERSION 19

#define JTVAE_E_NULL (-1)     /* required pointer is NULL            */
#define JTVAE_E_SHAPE (-2)    /* negative / inconsistent sizes       */
#define JTVAE_E_ALIGN (-3)    /* ld or base address not 16-B aligned */
#define JTVAE_E_RANGE (-4)    /* enum / mode out of range            */
#define JTVAE_E_WORKSPACE (-5)/* workspace too small                 */

/* activation codes for jtvae_gemm_nt */
#define JTVAE_ACT_NONE 0
#define JTVAE_ACT_RELU 1
#define JTVAE_ACT_SIGMOID 2
#define JTVAE_ACT_TANH 3

/* epilogue modes for jtvae_gemm_nt */
#define JTVAE_EPI_PLAIN 0     /* C = act(acc + bias + add)                                   */
#define JTVAE_EPI_GRU_BLEND 1 /* p = tanh(acc+add); C = p; out2 = (1-z)*s + z*p, 0 on zero_row */
#define JTVAE_EPI_MUL_ACC 2   /* C = acc (+add); out2 += C * aux0                            */

typedef struct jtvae_epilogue {
  const float* bias;      /* [N] added to every row, or NULL                               */
  const float* add;       /* addend rows, or NULL; may alias C (in-place accumulate)       */
  int32_t ld_add;
  const int32_t* add_idx; /* NULL: addend row = r; else addend row = add_idx[r] (table),   */
                          /* r = id of the output row (see c_rows)                         */
  int32_t act;            /* JTVAE_ACT_*  (PLAIN mode only)                                */
  float* pre_out;         /* PLAIN: optional copy of the pre-activation value, or NULL     */
  int32_t ld_pre;
  int32_t mode;           /* JTVAE_EPI_*                                                   */
  const float* aux0;      /* GRU_BLEND: z rows;  MUL_ACC: multiplier rows                  */
  const float* aux1;      /* GRU_BLEND: s rows                                             */
  float* out2;            /* GRU_BLEND: new h;   MUL_ACC: accumulator rows (+=)            */
  int32_t ld_aux;         /* leading dimension of aux0, aux1, out2                         */
  const int32_t* a_rows;  /* NULL: A row of product row i is i; else a_rows[i] (gather)    */
  const int32_t* c_rows;  /* NULL: product row i is output row i; else row c_rows[i] of C, */
                          /* pre_out, add, aux0, aux1, out2 (scatter by message id)        */
  int32_t zero_row;       /* GRU_BLEND: output row id whose out2 is forced to 0, -1 = none */
  const int32_t* aux_sf;  /* MUL_ACC: if set, aux0 row r is read from slab                 */
  int32_t aux_t;          /*   tau = min(aux_t, aux_sf[r] - 1) of aux0 (slab 0 == zero),   */
  size_t aux_slab_stride; /*   slabs aux_slab_stride floats apart                          */
} jtvae_epilogue_t;

int jtvae_abi_version(void);

/* Number of kernels jtvae_gemm_nt, jtvae_gemm_nt_pair and jtvae_gru_update have launched from the calling host
 * thread so far.  A pair or a fused update that does not fit its one-launch form (row counts, alignment) runs as
 * two launches; per-launch measurements read this counter before and after a call.  No counterpart in the
 * reference. */
uint64_t jtvae_nt_launch_count(void);
const char* jtvae_error_string(int code);

/* ---- dense contractions on the matrix cores ------------------------------------------- */

/* Form of the large products.  jtvae_gemm_nt with more than ~10 000 rows and jtvae_gemm_tn over at least 4096 rows
 * (inner extent >= 64, 16-byte aligned operands) run on the bf16 matrix instruction with every fp32 operand split
 * into three bf16 parts and the six largest cross products accumulated in fp32 (v_mfma_f32_32x32x16_bf16): the
 * result is as close to the exact product as the fp32 fma chain of v_mfma_f32_32x32x2_f32 (measured against fp64:
 * 3e-7 against 4e-7 of the result's rms at K = 450), but not bit-identical to it.  Everything smaller stays on the
 * fp32 instruction.  on = 0 keeps every product on the fp32 instruction, on = 1 restores the default, on < 0 only
 * queries; returns the setting in force before the call.  Process-wide; the environment variable
 * JTVAE_SPLIT_BF16=0 sets the initial state.  No counterpart in the reference (torch.nn.Linear picks its own
 * GEMM kernel by size as well). */
int jtvae_set_split_products(int on);

/* Row count up to which the products issued by the CALLING THREAD take the one-tile-per-block kernels (whole inner
 * extent in 115 KB of LDS, one memory round trip: the fastest form of a launch that has the GPU to itself); rows < 0
 * restores the default (512).  Those kernels need a CU to themselves, so a launch that runs beside a large product
 * of another stream waits until CUs drain; the 128 x 32 tiling (23 KB of LDS, <= 80 registers) it falls back to with
 * rows = 0 finds room beside three resident blocks of the large products at once.  Returns the previous setting
 * (-1 = default).  No counterpart in the reference. */
int jtvae_set_small_tile_rows(int rows);

/* C[M,N] = epilogue(A[M,K] . W[N,K]^T).  Replaces every nn.Linear call on the path:
 * fast_jtnn/mpn.py:87,93,99; fast_jtnn/jtnn_enc.py:33,115,117-118,124;
 * fast_jtnn/jtmpn.py:89,96,103 (the torch.cat in front of W_o/W_z/W_h/outputNN is replaced
 * by splitting the weight and chaining two calls through `add`).  A, W may have any ld and
 * alignment (unaligned operands take a scalar-load path); C, add, aux need no alignment. */
int jtvae_gemm_nt(const float* A, int32_t lda, const float* W, int32_t ldw, float* C, int32_t ldc,
                  int32_t M, int32_t N, int32_t K, const jtvae_epilogue_t* epi, void* stream);

/* Two independent NT products (neither reads what the other writes) issued together: when both are
 * small or mid-size and carry the epilogue pair of the tree GRU's backward pass (first PLAIN, second
 * MUL_ACC: ds += dz.Wz and dg = dp.Wh with its reset-gate accumulation, the autograd of
 * fast_jtnn/jtnn_enc.py:115,124) they run as ONE launch and share the otherwise idle CUs; any other
 * combination runs as two launches.  Results are those of two jtvae_gemm_nt calls. */
typedef struct jtvae_gemm_nt_problem {
  const float* A;
  int32_t lda;
  const float* W;
  int32_t ldw;
  float* C;
  int32_t ldc;
  int32_t M, N, K;
  const jtvae_epilogue_t* epi; /* NULL = plain product */
} jtvae_gemm_nt_problem_t;
int jtvae_gemm_nt_pair(const jtvae_gemm_nt_problem_t* p0, const jtvae_gemm_nt_problem_t* p1, void* stream);

/* dW[N,K] (+)= sum_m X[xr(m), 0:N]^T . Y[yr(m), 0:K]  - the weight gradient of C = Y.W^T given
 * X = dC, reduced over M in fixed order (split-M slabs + ordered slab sum: bitwise reproducible, no
 * atomics).  x_rows / y_rows (int32 [M], or NULL = identity) pick the row of X and of Y for every
 * reduction index: the tree GRU contracts compactly stored gradient rows against saved state rows
 * scattered over per-iteration slabs.  Replaces autograd's addmm backward for every Linear on the
 * path.  dbias ([N], or NULL): also dbias[n] (+)= sum_m X[xr(m), n], the gradient of the Linear's
 * bias, from a virtual column of ones appended to Y (it rides in the padding of the last column
 * tile, so it costs no extra pass over X).
 * workspace: at least jtvae_gemm_tn_workspace_bytes(M,N,K) bytes of device memory. */
size_t jtvae_gemm_tn_workspace_bytes(int32_t M, int32_t N, int32_t K);
int jtvae_gemm_tn(const float* X, int32_t ldx, const int32_t* x_rows, const float* Y, int32_t ldy,
                  const int32_t* y_rows, float* dW, int32_t lddw, float* dbias, int32_t M, int32_t N,
                  int32_t K, int32_t accumulate, void* workspace, size_t workspace_bytes, void* stream);

/* jtvae_gemm_tn without a workspace argument and without any allocation: the slab partials go to a
 * fixed array in device memory that the library holds on every GPU it runs on until the process
 * ends (jtvae_gemm_tn_static_max_bytes() = 64 MiB: enough for every N, K <= 450 product at any M).
 * It has no counterpart in the reference; it is for callers that cannot allocate (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same slab plan as
 * jtvae_gemm_tn - results are bit-identical to it - while that plan's partials fit the array (up to ~0.7 M
 * rows for N = K = 450); longer products keep one round of longer slabs here, jtvae_gemm_tn runs several rounds
 * of 8192-row slabs, and the two agree to fp32 rounding.  A product whose partials need more than the
 * array holds is rejected with JTVAE_E_WORKSPACE before any launch.  All calls on one GPU share the
 * array: calls that overlap on one GPU (two streams, two threads) race; issue them on one stream.
 * The array used is the one of the GPU that owns `stream` (of the current device for the NULL stream); the
 * operands must live on that GPU.  The first call for a device resolves the array's address
 * (hipGetSymbolAddress), later calls only launch.  Errors: JTVAE_E_* (negative) as everywhere, or the positive
 * hipError_t of the address lookup / launch. */
size_t jtvae_gemm_tn_static_max_bytes(void);
int jtvae_gemm_tn_static(const float* X, int32_t ldx, const int32_t* x_rows, const float* Y, int32_t ldy,
                         const int32_t* y_rows, float* dW, int32_t lddw, float* dbias, int32_t M, int32_t N,
                         int32_t K, int32_t accumulate, void* stream);

/* Up to JTVAE_PACK_MAX copies / transposes in one launch: the weights of a module laid out as
 * 16-byte-row GEMM operands.  transpose == 0: dst[r, c] = src[r, c]; transpose != 0:
 * dst[c, r] = src[r, c]; r < rows, c < cols; the remaining columns of every destination row up
 * to ld_dst are written as zero.  `items` is a host array, read during the call.  Replaces the
 * implicit layout handling of nn.Linear (weight slices of torch.cat inputs, fast_jtnn/mpn.py:98,
 * jtnn_enc.py:114,123) and the weight transposes of autograd's addmm backward. */
#define JTVAE_PACK_MAX 16
typedef struct jtvae_pack_item {
  const float* src;
  float* dst;
  int32_t ld_src, ld_dst;
  int32_t rows, cols; /* extent of the source block */
  int32_t transpose;
} jtvae_pack_item_t;
int jtvae_pack_batch(const jtvae_pack_item_t* items, int32_t n_items, void* stream);

/* dst[r, 0:cols] = src[r, 0:cols]; dst columns [cols, ld_dst) = 0.  Any alignment.  Moves rows
 * between torch's dense [rows, H] tensors and the padded row buffers. */
int jtvae_copy_rows(const float* src, int32_t ld_src, float* dst, int32_t ld_dst, int32_t rows,
                    int32_t cols, void* stream);

/* dst[r, 0:cols] = (float)src[r, 0:cols] (src dense [rows, cols] bytes); dst columns [cols, ld_dst) = 0.
 * The device half of shipping the one-hot feature rows (fast_jtnn/mpn.py:18-31, moved by create_var at
 * mpn.py:82-85) as bytes; see jtvae_prep_features_to_u8. */
int jtvae_widen_u8_rows(const uint8_t* src, float* dst, int32_t ld_dst, int32_t rows, int32_t cols,
                        void* stream);

/* ---- linear layers on one-hot feature rows ------------------------------------------------ */

/* out[r, 0:H] = bias + sum_j feats[r, j] * Wt[j, 0:H]  (terms in increasing j, zero entries skipped), columns
 * [H, ld) = 0.  feats: [n_rows, F] bytes, dense; Wt: the transposed weight [F, ldw] (jtvae_pack_batch with
 * transpose != 0); bias: [H] or NULL.  Replaces the Linear calls whose input is a feature matrix of one-hot blocks:
 * W_i on fbonds (fast_jtnn/mpn.py:87, jtmpn.py:89) and the atom-feature columns of W_o (mpn.py:98-99,
 * jtmpn.py:102-103).  A row kernel bound by the rows it writes; the table must fit in LDS
 * (F * ldw * 4 + 4 KiB <= 150 KiB, else JTVAE_E_WORKSPACE: use jtvae_gemm_nt on the widened rows). */
int jtvae_onehot_linear(const uint8_t* feats, int32_t n_rows, int32_t F, const float* Wt, int32_t ldw,
                        const float* bias, float* out, int32_t ld, int32_t H, void* stream);

/* dW[n, j] (+)= sum_r feats[r, j] * dOut[r, n]: the weight gradient of the layer above in nn.Linear layout
 * ([H, lddw >= F]); autograd's addmm backward for mpn.py:87,99 / jtmpn.py:89,103.  Rows are summed in fixed order
 * (per-block tables in LDS, then the blocks in order): bitwise reproducible.  workspace: at least
 * jtvae_onehot_linear_bwd_workspace_bytes(n_rows, F, ld) bytes of device memory. */
size_t jtvae_onehot_linear_bwd_workspace_bytes(int32_t n_rows, int32_t F, int32_t ld);
int jtvae_onehot_linear_bwd(const uint8_t* feats, int32_t n_rows, int32_t F, const float* dOut, int32_t ld, int32_t H,
                            float* dW, int32_t lddw, int32_t accumulate, void* workspace, size_t workspace_bytes,
                            void* stream);

/* ---- neighbour gather-sum --------------------------------------------------------------- */

/* out[i] = sum_{e in [rowptr[i], rowptr[i+1])} src(col[e])        for i < n_rows
 *   src(j) = src0[j]            if j <  n_src0
 *          = src1[j - n_src0]   otherwise          (src1 == NULL: single source)
 *   relu_src != 0: rows of the last source (src1 if given, else src0) are pre-activations and enter the
 *   sum as max(row, 0) (message = relu(binput), mpn.py:88 / jtmpn.py:90, without storing the activated copy)
 *   then, in this order:  if relu_mask:  out[i] *= (relu_mask[i] > 0)
 *                         if accumulate: out[i] += previous out[i]
 *                         if acc2:       acc2[i] += out[i]
 * Replaces index_select_ND(...).sum(dim=1) (fast_jtnn/nnutils.py:13-18 with mpn.py:91-92,
 * 96-97; jtnn_enc.py:31-32; jtmpn.py:93-95, 99-101 - the two-source form stands in for the
 * torch.cat([tree_message, graph_message]) copy) and, with the transposed CSR, its autograd
 * backward (index_select backward = scatter-add) fused with the ReLU mask of mpn.py:94.
 * All row buffers share `ld`. */
int jtvae_csr_gather_sum(float* out, const float* src0, int32_t n_src0, const float* src1,
                         const int32_t* rowptr, const int32_t* col, int32_t n_rows, int32_t H,
                         int32_t ld, const float* relu_mask, int32_t accumulate, float* acc2,
                         int32_t relu_src, void* stream);

/* Per-segment mean readout: out[b] = (1/len_b) * sum rows[start_b : start_b+len_b]
 * (fast_jtnn/mpn.py:103-107, fast_jtnn/jtmpn.py:105-110).  seg = int32 [n_seg,2] (start,len). */
int jtvae_segment_mean(float* out, const float* rows, const int32_t* seg, int32_t n_seg, int32_t H,
                       int32_t ld, void* stream);

/* Backward of the readout fused with the ReLU in front of it (mpn.py:99):
 * drows[a] = dout[seg_of[a]] / len(seg_of[a]) * (act[a] > 0). */
int jtvae_segment_mean_bwd(float* drows, const float* dout, const float* act, const int32_t* seg,
                           const int32_t* seg_of, int32_t n_rows, int32_t H, int32_t ld, void* stream);

/* ---- GraphGRU (fast_jtnn/jtnn_enc.py:107-128) -------------------------------------------
 *
 * Active rows.  The reference updates every message in each of the `depth` iterations.  The update
 * is synchronous, h_{t+1}(m) = F(h_t of the messages m reads), so a message that reads nothing is
 * constant from iteration 1 on and a message whose inputs are constant from iteration k on is
 * constant from k+1 on ("stable_from", sf[m]; on a tree: the height of the sub-tree behind the
 * message).  Re-computing a constant row gives the bit-identical value, so the entry points below
 * take a LIST of row ids to update (`rows`, NULL = all n rows in order) and, for state kept per
 * iteration in slabs `slab_stride` floats apart, read the value row j had at iteration t from slab
 * min(t, sf[j]) (sf == NULL: single buffer, no slabs).  With rows == NULL and sf[j] >= depth this is
 * the reference's dense schedule. */

/* First iteration, where h == 0: z = sigmoid(Pz[wid]), p = tanh(Ph[wid]), h1 = z*p, row pad_row -> 0.
 * Pz/Ph are per-word (or per-node) tables of W_z[:, :H].x + b_z and W_h[:, :H].x + b_h.
 * z, p may be NULL (inference). */
int jtvae_gru_iter0(const float* Pz, const float* Ph, const int32_t* wid, float* z, float* p,
                    float* h1, int32_t n_mess, int32_t H, int32_t ld, int32_t pad_row, void* stream);

/* One pass over the incoming messages of every active message m (jtnn_enc.py:112-113, 117-122):
 *   s[m] = sum_e h[j_e]
 *   g[m] = sum_e sigmoid(Pr[wid[m]] + Uh[j_e]) * h[j_e]        Uh = U_r h + b, once per message
 *   t[m] = sum_e r_e (1 - r_e) h[j_e]                         (t may be NULL: inference)
 * h, Uh: slab 0 base pointers (see above).  s, g, t: output buffers indexed by message id.
 * The [NM,K,H] tensors h_nei, r, gated_h of the reference are never materialised. */
int jtvae_gru_gate_gather(const float* h, const float* Uh, size_t slab_stride, int32_t t_iter,
                          const int32_t* sf, const float* Pr, const int32_t* wid, const int32_t* rowptr,
                          const int32_t* col, const int32_t* rows, int32_t n_rows, float* s, float* g,
                          float* t, int32_t H, int32_t ld, void* stream);

/* Update of the active rows of one iteration in one launch (jtnn_enc.py:114-115,123-126; the decoder's
 * nnutils.py:59-60,68-70):  z = sigmoid(S . Wz^T + Pz[wid]),  p = tanh(G . Wh^T + Ph[wid]),
 * h' = (1 - z) s + z p, row zero_row of h' forced to 0.  S, G: the two neighbour sums of
 * jtvae_gru_gate_gather (rows by message id, leading dimension ld); Wz = W_z[:, H:], Wh = W_h[:, H:]
 * ([H, ldw]); Pz, Ph: the per-word tables; rows: message ids to update (NULL = 0 .. n_rows-1).  Z, P, Hout
 * by message id.  Results are those of two jtvae_gemm_nt calls (sigmoid epilogue, then GRU_BLEND). */
int jtvae_gru_update(const float* S, const float* G, int32_t ld, const float* Wz, const float* Wh, int32_t ldw,
                     const float* Pz, const float* Ph, int32_t ldp, const int32_t* wid, const int32_t* rows,
                     int32_t n_rows, float* Z, float* P, float* Hout, int32_t H, int32_t zero_row, void* stream);

/* Backward of the blend h' = ((1-z) s + z p) * mask (jtnn_enc.py:115,124-126) for the rows that
 * receive gradient at iteration t_iter:
 *   dzpre = dh (p - s) z (1-z);  dppre = dh z (1-p^2);  ds = dh (1-z);  dh[pad_row] taken as 0.
 * z, p, s: slab 0 bases; row m is read from slab min(t_iter, sf[m]-1), s of slab 0 is zero.
 * dzpre / dppre are written COMPACTLY (row i of the launch), ds by message id. */
int jtvae_gru_bwd_pointwise(const float* dh, const float* z, const float* p, const float* s,
                            size_t slab_stride, int32_t t_iter, const int32_t* sf, const int32_t* rows,
                            int32_t n_rows, float* dzpre, float* dppre, float* ds, int32_t H, int32_t ld,
                            int32_t pad_row, void* stream);

/* Backward of jtvae_gru_gate_gather with respect to h and Uh, over the TRANSPOSED graph
 * (trowptr/tcol: for message j, the messages m that read it), r recomputed:
 *   dUh[i]   = h[j] * sum_e dg[m_e] r_e (1 - r_e)             written compactly (row i of the launch)
 *   dhdir[j] = sum_e ( ds[m_e] + dg[m_e] r_e ) (+ dh_ext[j]) r_e = sigmoid(Pr[wid[m_e]] + Uh[j])
 * Readers m_e whose bit t_iter of active[m_e] is clear received no gradient and are skipped
 * (active == NULL: none skipped).  dh_ext (or NULL): gradient rows by message id that reach h from
 * outside the recurrence (the readout, the candidate encoder); it may alias dhdir's target of the
 * following product, it is only read here. */
int jtvae_gru_bwd_gather(const float* h, const float* Uh, size_t slab_stride, int32_t t_iter,
                         const int32_t* sf, const uint64_t* active, const float* Pr, const int32_t* wid,
                         const float* dg, const float* ds, const int32_t* trowptr, const int32_t* tcol,
                         const int32_t* rows, int32_t n_rows, const float* dh_ext, float* dUh, float* dhdir,
                         int32_t H, int32_t ld, void* stream);

/* One level of the backward recurrence from the gathered dU_r h rows to the pre-activation gradients, in one
 * launch where the shapes allow it: dh[rows[i]] = du[i] . UrT^T + dhdir[rows[i]] (the jtvae_gemm_nt that follows
 * jtvae_gru_bwd_gather), and from that value, in the epilogue, what jtvae_gru_bwd_pointwise computes: dzpre[i],
 * dppre[i] (compact) and ds[rows[i]].  Same arguments and the same bits as the two calls; rows beyond 10 000 or
 * unaligned operands run as those two launches.  UrT: [H, ldw] (U_r transposed). */
int jtvae_gru_bwd_dh(const float* du, const float* UrT, int32_t ldw, const float* dhdir, float* dh,
                     const int32_t* rows, int32_t n_rows, const float* z, const float* p, const float* s,
                     size_t slab_stride, int32_t t_iter, const int32_t* sf, float* dzpre, float* dppre, float* ds,
                     int32_t H, int32_t ld, int32_t pad_row, void* stream);

/* ---- the graph message passing network, one host call per pass ---------------------------------- */

/* MPN.forward / JTMPN.forward (fast_jtnn/mpn.py:81-108, fast_jtnn/jtmpn.py:81-111) on prepared inputs: float
 * feature rows fbonds [n_bonds, ld_fb] / fatoms [n_atoms, ld_fa], the weight blocks W_i [H, ld_wi], W_h [H, ld],
 * W_o[:, :fa_dim] [H, ld_woa], W_o[:, fa_dim:] [H, ld], bias bo [H], the CSR forms of bgraph / agraph (ids below
 * n_tree are rows of `tree`, the rest bond rows; tree = NULL, n_tree = 0 for MPN) and the segment table.  msg / nei
 * are HOST arrays of device row buffers [n_bonds, ld]: msg[0] receives binput, msg[i] the messages after iteration
 * i (depth entries), nei[i-1] the neighbour sums of iteration i (depth - 1 entries); a caller that keeps nothing for
 * a backward pass may let nei entries alias each other and msg[i] alias msg[i-2].  anei, ah: [n_atoms, ld]; out:
 * [n_seg, ld].  Issues jtvae_gemm_nt, (jtvae_csr_gather_sum, jtvae_gemm_nt) x (depth - 1), jtvae_csr_gather_sum,
 * jtvae_gemm_nt x 2, jtvae_segment_mean: bit-identical to issuing them one by one. */
typedef struct jtvae_mpn_fwd {
  const float* fbonds;
  const float* fatoms;
  const float *Wi, *Wh, *Woa, *Woh, *bo;
  const float* tree;
  const int32_t *b_rowptr, *b_col, *a_rowptr, *a_col;
  const int32_t* seg;
  float* const* msg;
  float* const* nei;
  float *anei, *ah, *out;
  int32_t ld_fb, ld_fa, ld_wi, ld_woa;
  int32_t n_tree, n_bonds, n_atoms, n_seg, fb_dim, fa_dim, H, ld, depth;
} jtvae_mpn_fwd_t;
int jtvae_graph_mpn_forward(const jtvae_mpn_fwd_t* args, void* stream);

/* The matching backward pass (autograd of the same lines): dout [n_seg, ld] in; dW_i [H, ld_dwi], dW_h [H, H],
 * dW_o [H, ld_dwo] (atom columns first), db_o [H] and, with tree rows, dtree [n_tree, ld] out.  msg, nei, anei, ah:
 * what the forward call left; WhT / WohT: the transposed weight blocks [H, ld]; at_* / bt_*: transposed CSR of
 * agraph / bgraph over (tree rows | bond rows); seg_of: segment of every atom.  Scratch by the caller: dpre_o, danei
 * [n_atoms, ld]; dpre, dnei, dbinput [n_bonds, ld]; workspace of at least the largest
 * jtvae_gemm_tn_workspace_bytes among the four weight-gradient products.  Bit-identical to the launch-by-launch
 * form. */
typedef struct jtvae_mpn_bwd {
  const float* dout;
  const float* fbonds;
  const float* fatoms;
  const float *WhT, *WohT;
  float* const* msg;
  float* const* nei;
  const float *anei, *ah;
  const int32_t *at_rowptr, *at_col, *bt_rowptr, *bt_col;
  const int32_t *seg, *seg_of;
  float *dpre_o, *danei, *dpre, *dnei, *dbinput;
  float *dW_i, *dW_h, *dW_o, *db_o, *dtree;
  void* workspace;
  size_t workspace_bytes;
  int32_t ld_fb, ld_fa, ld_dwi, ld_dwo;
  int32_t n_tree, n_bonds, n_atoms, n_seg, fb_dim, fa_dim, H, ld, depth;
} jtvae_mpn_bwd_t;
int jtvae_graph_mpn_backward(const jtvae_mpn_bwd_t* args, void* stream);

/* ---- whole level sweeps of the compute-once schedule, one host call each ---------------------- */

/* Levels 1 .. n_levels of GraphGRU.forward (jtnn_enc.py:110-127) with every message computed once, in the level
 * in which it settles (level 0 is jtvae_gru_iter0).  All state buffers hold one row per message ([NM, ld]):
 * h (in/out), uh = U_r h + b, s, g, t (NULL: inference), z, p.  Per level L: f_rows[L] / f_n[L] = device row
 * list and count of the messages it updates, u_rows[L] / u_n[L] = the messages whose U_r product it needs first
 * (those that settled in the level before and are read by someone); the arrays of pointers and counts live in
 * HOST memory.  Pz / Pr / Ph: per-word tables [n_words, ld]; wid: table row of a message; rowptr / col: the
 * messages each message reads; Wzh = W_z[:, H:], Whh = W_h[:, H:], Ur: [H, ld]; bu: [H] or NULL.  Issues, per
 * level, jtvae_gemm_nt + jtvae_gru_gate_gather + jtvae_gru_update in that order: the result is bit-identical
 * to issuing them one by one.  Stops at the first level with f_n == 0. */
typedef struct jtvae_tree_fwd {
  float *h, *uh, *s, *g, *t, *z, *p;
  const int32_t* sf;
  const float *Pz, *Pr, *Ph;
  const int32_t* wid;
  const int32_t *rowptr, *col;
  const float *Wzh, *Whh, *Ur, *bu;
  int32_t H, ld, pad_row, n_levels;
  const int32_t* const* f_rows;
  const int32_t* f_n;
  const int32_t* const* u_rows;
  const int32_t* u_n;
} jtvae_tree_fwd_t;
int jtvae_tree_gru_forward(const jtvae_tree_fwd_t* args, void* stream);

/* The matching backward sweep (autograd of jtnn_enc.py:110-127 under the same schedule): every gradient-carrying
 * message once, in the reverse dependency levels of jtvae_prep_backward_once.  e_rows[L] / e_n[L]: device row
 * list and count of level L, n_dep[L]: its leading rows that read other messages, off_e[L] / off_j[L]: first
 * compact row of the level in dz, dp / in du (host arrays).  dh: gradient of the final h by message id (in/out),
 * dz, dp, du: compact outputs, ds, dg, dhdir: scratch by message id, a_r: accumulator by message id (zeroed by
 * the caller), T: the iteration count of the forward pass, WzhT / WhhT / UrT: the transposed weight blocks.
 * Issues, per level, jtvae_gru_bwd_gather + jtvae_gemm_nt (levels > 0), jtvae_gru_bwd_pointwise and
 * jtvae_gemm_nt_pair, bit-identical to issuing them one by one. */
typedef struct jtvae_tree_bwd {
  const float *h, *uh, *s, *t, *z, *p;
  const int32_t* sf;
  const uint64_t* active;
  const float* Pr;
  const int32_t* wid;
  const int32_t *trowptr, *tcol;
  const float *WzhT, *WhhT, *UrT;
  float *dh, *dz, *dp, *du, *ds, *dg, *dhdir, *a_r;
  int32_t T, H, ld, pad_row, n_levels;
  const int32_t* const* e_rows;
  const int32_t* e_n;
  const int32_t* n_dep;
  const int64_t* off_e;
  const int64_t* off_j;
} jtvae_tree_bwd_t;
int jtvae_tree_gru_backward(const jtvae_tree_bwd_t* args, void* stream);

/* JTNNEncoder.forward (fast_jtnn/jtnn_enc.py:20-42) around its level sweep as ONE host call: the embedding rows of
 * the words used, the four per-word tables (W_z / W_h / outputNN input halves with their biases, W_r), iteration 0,
 * the sweep (jtvae_tree_gru_forward on `sweep`, whose Pz / Pr / Ph / h / z / p / wid are the buffers used here), the
 * root sums and the output layer on the roots.  `side` (or NULL): a second stream for the reset-gate and output
 * tables, which the chain of levels reads only from level 1 on / after its last level; forked and joined inside.
 * train != 0: iteration 0 stores z and p.  crowded bit 0: the table products take the 128 x 32 tiling
 * (jtvae_set_small_tile_rows(0) for their duration), bit 1: so do the short levels of the sweep.  The same
 * launches in the same order as issuing them one by one: bit-identical. */
typedef struct jtvae_csr {
  const int32_t* rowptr;
  const int32_t* col;
  int32_t n_rows;
} jtvae_csr_t;
typedef struct jtvae_tree_enc_fwd {
  int32_t NM, B, NW, V, H, ld, crowded, train;
  const float* E;                          /* embedding as a row buffer [V, ld] */
  jtvae_csr_t word, root;                  /* word rows used (NW rows); messages into each root (B rows) */
  const float *Wzx, *Whx, *Wox, *Wr, *Woh; /* weight blocks [H, ld] */
  const float *bz, *bh, *bout;
  const int32_t* wid_root;                 /* table row of each root's word */
  const jtvae_tree_fwd_t* sweep;
  float *Eu, *Po;                          /* [NW, ld]: word rows (kept for the backward pass), output table */
  float *nsum, *tv;                        /* [B, ld]: root sums (kept), tree vectors */
} jtvae_tree_enc_fwd_t;
int jtvae_tree_encoder_forward(const jtvae_tree_enc_fwd_t* args, void* stream, void* side);

/* Everything of JTNNEncoder's backward pass that surrounds the level sweep (autograd of fast_jtnn/jtnn_enc.py:27-41
 * and of the per-word table products that replace its W_z / W_r / W_h / outputNN input halves), as ONE host call:
 * the root stage (ReLU mask, W_out gradient, gradient into the messages that reach a root), the sweep
 * (jtvae_tree_gru_backward on `sweep`), the three weight-gradient contractions of the recurrence, the four table
 * gradients (two keyed sums and one contraction each) and the embedding gradient.  `side` (or NULL) is a second
 * stream for the groups of launches nothing on the main chain waits for (root-stage weight gradients beside the
 * sweep, recurrence weight gradients beside the table gradients); the call forks and joins it with events of its
 * own, so on return everything is ordered on `stream`.  The same launches in the same order on the same data as
 * issuing them one by one: bit-identical.  A keyed sum is two gathers: `*_pieces` sums runs of source rows, `*_key`
 * sums the pieces of a table row.  crowded != 0: products of <= 512 rows take the 128 x 32 tiling for the duration
 * of the call (jtvae_set_small_tile_rows(0)).  ws[i] / ws_bytes[i]: one jtvae_gemm_tn workspace per contraction
 * (0: W_out h-half, 1: o table, 2: z h-half, 3: h h-half, 4: U_r, 5: z table, 6: h table, 7: r table). */
typedef struct jtvae_tree_enc_bwd {
  int32_t NM, B, NW, V, H, ld, crowded;
  const float *Eu, *nsum, *tv;          /* kept by the forward pass: word rows used, root sums, root outputs */
  const float* dtv;                     /* [B, ld] gradient of tree_vecs, or NULL (zero) */
  const float* dmess;                   /* gradient of messages (rows ld_dmess apart), or NULL */
  int32_t ld_dmess;
  const float *WohT, *WzxT, *WhxT, *WoxT, *WrT; /* transposed weight blocks [H, ld] */
  jtvae_csr_t ident, root_t, word_t;    /* identity over B; root sums transposed (NM rows); word rows -> vocabulary (V rows) */
  jtvae_csr_t o_pieces, o_key, zh_pieces, zh_key, r_pieces, r_key; /* keyed sums into the NW table rows */
  const jtvae_tree_bwd_t* sweep;        /* its dh / dz / dp / du / a_r are read and written here too */
  int32_t n_e;                          /* rows of sweep->dz and sweep->dp */
  const float *ss, *gs, *hs;            /* saved s, g, h by message id */
  const int32_t *zx, *zy, *ux, *uy;     /* row pairs of the recurrence's weight gradients */
  int32_t n_z, n_u;
  float *dpre_root, *dnsum, *part_side, *part_main, *dPo, *dPz, *dPh, *dPr, *dEu, *dE_p; /* scratch row buffers */
  void* ws[8];
  size_t ws_bytes[8];
  float *dWout, *dWz, *dWh;             /* [H, 2H] dense */
  float *dWr, *dUr;                     /* [H, H] dense */
  float *dbout, *dbz, *dbh, *dbu;       /* [H] */
  float* dE;                            /* [V, H] dense */
} jtvae_tree_enc_bwd_t;
int jtvae_tree_encoder_backward(const jtvae_tree_enc_bwd_t* args, void* stream, void* side);

/* out[i] = slab sel[i] of base, row i: collects the final value of every message
 * (sel[i] = min(depth, sf[i])). */
int jtvae_select_slab_rows(float* out, const float* base, size_t slab_stride, const int32_t* sel,
                           int32_t n_rows, int32_t H, int32_t ld, void* stream);

/* ---- bilinear assembly score (fast_jtnn/jtnn_vae.py:112-114) ---------------------------- */

/* score[c] = < q[owner[c]], cand[c] >  for c < n_cand (q = A_assm z_mol rows). */
int jtvae_row_dot(float* score, const float* q, const int32_t* owner, const float* cand,
                  int32_t n_cand, int32_t H, int32_t ld, void* stream);

/* Backward of jtvae_row_dot (autograd of the bmm of fast_jtnn/jtnn_vae.py:114): dcand[c] = dscore[c] * q[owner[c]]
 * and dq[b] = sum of dscore[c] * cand[c] over the candidates of owner b, which must be consecutive:
 * own_ptr (int32 [n_owner + 1]) delimits them, as datautils.py:150-161 lays them out (molecule by molecule). */
int jtvae_row_dot_bwd(const float* dscore, const float* q, const int32_t* owner, const float* cand,
                      const int32_t* own_ptr, int32_t n_cand, int32_t n_owner, float* dq, float* dcand, int32_t H,
                      int32_t ld, void* stream);

/* Candidate-assembly loss: segment s covers scores[seg_ptr[s] .. seg_ptr[s+1]) (the candidates of one
 * tree node), label_pos[s] is the index of the true candidate.  loss_seg[s] = cross entropy of the
 * segment, hit_seg[s] = 1 if the label has the highest score, dscores (or NULL) = softmax - onehot.
 * Replaces the per-node narrow + CrossEntropyLoss + .item() loop of fast_jtnn/jtnn_vae.py:116-135
 * (jtprop_vae.py:454-473). */
int jtvae_segment_ce(const float* scores, const int32_t* seg_ptr, const int32_t* label_pos, int32_t n_seg,
                     float* loss_seg, int32_t* hit_seg, float* dscores, void* stream);

/* ---- host-side index preparation (host memory in and out; no GPU work, no stream) ---------- */

/* A zero-padded int64 neighbour table [rows, K] (agraph / bgraph of fast_jtnn/mpn.py:148-159 and
 * jtmpn.py:170-181, node_graph / mess_graph of jtnn_enc.py:78-90; 0 = no neighbour) as int32 CSR
 * without the padding slots, slot order kept.  rowptr: [rows+1]; col: capacity col_capacity
 * (rows*K always suffices); *nnz = entries written.  Values outside [0, n_src): JTVAE_E_RANGE. */
int jtvae_prep_padded_to_csr(const int64_t* table, int64_t rows, int32_t K, int64_t n_src,
                             int32_t* rowptr, int32_t* col, int64_t col_capacity, int64_t* nnz);

/* Transposed CSR: for every source j the rows that read it, in increasing (row, slot) order.  It
 * turns the scatter-add of autograd's index_select backward into a gather.  trowptr: [n_src+1];
 * tcol: [nnz]. */
int jtvae_prep_transpose_csr(const int32_t* rowptr, const int32_t* col, int64_t n_rows, int64_t n_src,
                             int32_t* trowptr, int32_t* tcol);

/* sf[m]: iteration from which message m of GraphGRU (jtnn_enc.py:107-128) no longer changes: 1 if
 * it reads no message, else 1 + max over the messages it reads, capped at max(T, 1). */
int jtvae_prep_stable_from(const int32_t* rowptr, const int32_t* col, int64_t n, int32_t T, int32_t* sf);

/* out[i] = (uint8_t)x[i] for i < n; *exact = 1 when every x[i] is a whole number in [0, 255] (the
 * conversion loses nothing), else 0 and out is incomplete.  Replaces nothing in the reference, which moves
 * the float rows as they are (nnutils.py:6-10 from mpn.py:82-85): it quarters the bytes of that move. */
int jtvae_prep_features_to_u8(const float* x, int64_t n, uint8_t* out, int32_t* exact);

/* Stable grouping by key (messages by the word of their source node, for the per-word table
 * gradients): order = item ids sorted by key, ties in input order; keyptr: [n_keys+1]. */
int jtvae_prep_group_by_key(const int32_t* keys, int64_t n, int64_t n_keys, int32_t* keyptr,
                            int32_t* order);

/* Backward plan of the compute-once GraphGRU schedule (the autograd backward of fast_jtnn/jtnn_enc.py:107-128
 * restated so that every gradient-carrying message is visited once).  rowptr / col: the messages each message
 * reads; seeds: messages with a gradient from outside the recurrence (n_seeds < 0: all of them).  Out: act [n]
 * (1 = carries gradient: a seed, or read by a message that does); order: those messages by (level, rows that read
 * something first, id), where level = 0 if no such message reads it, else 1 + the largest level among those
 * readers; lev_off [levels+1]; per level n_dep (rows that read something), deg_sum (entries they read), tdeg_sum
 * (readers among the visited messages).  Arrays need room for n (+1) entries.  counts[2] = {messages, levels}.
 * JTVAE_E_RANGE: an index is out of range or the dependencies have a cycle. */
int jtvae_prep_backward_once(const int32_t* rowptr, const int32_t* col, int64_t n, const int32_t* seeds,
                             int64_t n_seeds, uint8_t* act, int32_t* order, int32_t* lev_off, int32_t* n_dep,
                             int64_t* deg_sum, int64_t* tdeg_sum, int64_t* counts);

/* Forward plan of the compute-once GraphGRU schedule (fast_jtnn/jtnn_enc.py:111-126 with every message computed
 * in the iteration in which it settles).  sf_true: uncapped settling iteration of every message (jtvae_prep_stable_from
 * with a large T, or jtvae_tensorize_trees).  Out: sf [n] = min(sf_true, max(T, 1)); order [n]: messages by sf, ties in
 * id order; keyptr [max(T,1) + 2]: group k is order[keyptr[k] .. keyptr[k+1]); u_rows: the members of order that
 * some message reads, u_ptr [max(T,1) + 2] their group starts; nnz_ptr [max(T,1) + 2]: entries read by the groups
 * before k. */
int jtvae_prep_forward_once(const int32_t* rowptr, const int32_t* col, int64_t n, const int32_t* sf_true, int32_t T,
                            int32_t* sf, int32_t* order, int32_t* keyptr, int32_t* u_rows, int32_t* u_ptr,
                            int64_t* nnz_ptr);

/* Index form of the teacher-forced decoder traces of a batch (fast_jtnn/jtnn_dec.py:59-167, dfs :299-305),
 * from the junction-tree tables: directed edge m runs src[m] -> dst[m] (ids 1 .. n_mess-1 in the numbering of
 * fast_jtnn/jtnn_enc.py:58-62; 0 = padding message), wid = word of every node, roots = first node of every
 * tree.  Out: dep_rp [n_mess+1] / dep_col: the edges each edge's hidden row is computed from (neighbours so
 * far minus the target, creation order); per step row (ordered step by step, trees in batch order, the final
 * stop at every root last): stop_rp / stop_col (all edges that have reached the current node), stop_word,
 * stop_ctx (tree), stop_target (1 = down, 0 = back); pred_mess [n_down] (edge whose row predicts the child),
 * pred_ctx / pred_target [n_trees + n_down] (roots first).  Row arrays need room for n_mess - 1 + n_trees
 * entries; dep_cap / stop_cap = capacity of dep_col / stop_col.  counts[4] = {steps, n_down, nnz(dep),
 * nnz(stop)}.  JTVAE_E_RANGE: an index is out of range or the edges do not form trees. */
int jtvae_prep_decoder_plan(const int32_t* src, const int32_t* dst, int64_t n_mess, const int32_t* wid,
                            int64_t n_nodes, const int32_t* roots, int64_t n_trees, int32_t* dep_rp,
                            int32_t* dep_col, int64_t dep_cap, int32_t* stop_rp, int32_t* stop_col,
                            int64_t stop_cap, int32_t* stop_word, int32_t* stop_ctx, float* stop_target,
                            int32_t* pred_mess, int32_t* pred_ctx, int32_t* pred_target, int64_t* counts);

/* ---- host-side tensorizers: abstract graphs -> the reference's holder tables ----------------
 *
 * The reference builds its holders per batch in Python (fast_jtnn/datautils.py:140-163 calling
 * JTNNEncoder.tensorize, MPN.tensorize and JTMPN.tensorize).  Everything after the RDKit calls is index
 * bookkeeping over small graphs; the routines below do it natively from flat arrays, for a selection `ids`
 * (any order, repeats allowed; NULL = 0 .. n-1) of the graphs of a set, split over up to n_threads host threads
 * (molecules are independent).  Two output forms, either may be NULL:
 *   - `tables`: the reference's own layout, bit for bit (float feature rows, zero-padded int64 neighbour tables);
 *   - `csr`:    what the kernels consume (byte feature rows, int32 CSR without padding slots and its transpose,
 *               segment tables), so that the padded tables never exist.
 * Host memory in and out, no GPU work, no stream.  JTVAE_E_RANGE: an atom / node index is out of range or a
 * neighbour list overflows max_nb (the reference raises IndexError there). */

/* Small graphs stored back to back: molecules for MPN.tensorize (fast_jtnn/mpn.py:110-161) or attachment
 * candidates for JTMPN.tensorize (fast_jtnn/jtmpn.py:113-183).  Feature rows are the one-hot blocks of
 * mpn.py:46-68 / jtmpn.py:48-68 as bytes. */
typedef struct jtvae_graph_set {
  int64_t n_graphs;
  const int64_t* atom_ptr;   /* [n_graphs + 1] first atom of every graph                              */
  const int64_t* bond_ptr;   /* [n_graphs + 1] first (undirected) bond of every graph                 */
  const uint8_t* atom_feats; /* [atoms, atom_fdim]                                                    */
  const int32_t* bonds;      /* [bonds, 2] begin atom, end atom: positions inside the graph           */
  const uint8_t* bond_feats; /* [bonds, bond_fdim]                                                    */
  const int32_t* atom_nid;   /* candidates: [atoms] the atom-map number of jtmpn.py:140 (0 = none,    */
                             /* else 1-based position of a node in the owning tree); NULL: molecules  */
  int32_t atom_fdim, bond_fdim;
} jtvae_graph_set_t;

/* The tree messages of a batch, for the candidate graphs' extra in-edges (the mess_dict of
 * fast_jtnn/jtnn_enc.py:58-62 as arrays): message m runs src[m] -> dst[m] (batch-global node ids, m = 0 is the
 * padding message); node_base[i] = batch-global id of the first node of the tree that selected graph i belongs to. */
typedef struct jtvae_mess_index {
  int64_t n_mess, n_nodes;
  const int32_t* src;
  const int32_t* dst;
  const int64_t* node_base; /* [n] */
} jtvae_mess_index_t;

typedef struct jtvae_graph_tables { /* (fatoms, fbonds, agraph, bgraph, scope) of mpn.py:161 / jtmpn.py:183 */
  float* fatoms;   /* [atoms, atom_fdim]                                        */
  float* fbonds;   /* [bond rows, atom_fdim + bond_fdim]                        */
  int64_t* agraph; /* [atoms, max_nb]                                           */
  int64_t* bgraph; /* [bond rows, max_nb]                                       */
  int64_t* scope;  /* [n, 2] (first atom, atoms)                                */
} jtvae_graph_tables_t;

typedef struct jtvae_graph_csr {
  uint8_t* fatoms;     /* [atoms, atom_fdim]                                                      */
  uint8_t* fbonds;     /* [bond rows, atom_fdim + bond_fdim]                                      */
  int32_t* a_rowptr;   /* [atoms + 1]      incoming bonds (and tree messages) of every atom       */
  int32_t* a_col;      /* capacity a_cap                                                          */
  int32_t* b_rowptr;   /* [bond rows + 1]  the bonds a directed bond reads                        */
  int32_t* b_col;      /* capacity b_cap                                                          */
  int64_t a_cap, b_cap;
  int32_t *at_rowptr, *at_col; /* transposes over the n_src = n_mess + bond rows sources (n_mess = 0 for */
  int32_t *bt_rowptr, *bt_col; /* molecules): [n_src + 1], capacity a_cap / b_cap; NULL = not wanted     */
  int32_t* seg;        /* [n, 2] (first atom, atoms)                                              */
  int32_t* seg_of;     /* [atoms] graph of every atom                                             */
  int64_t nnz[4];      /* out: entries of a, of b, and how many of each point at tree messages    */
} jtvae_graph_csr_t;

/* counts[4] = {atoms, undirected bonds, a_cap, b_cap}: sizes of what jtvae_tensorize_graphs will write for the
 * same selection (bond rows = 2 * bonds, + 1 padding row for molecules; the caps are tight upper bounds). */
int jtvae_tensorize_graph_counts(const jtvae_graph_set_t* set, const int64_t* ids, int64_t n,
                                 const jtvae_mess_index_t* mess, int64_t* counts);

/* mess == NULL: MPN.tensorize (directed bond ids from 1, row 0 of fbonds / bgraph is the padding row,
 * mpn.py:112-114); else JTMPN.tensorize (bond ids from n_mess, bond rows from 0, tree-message in-edges for
 * bonds whose atoms carry different node ids, jtmpn.py:140-162). */
int jtvae_tensorize_graphs(const jtvae_graph_set_t* set, const int64_t* ids, int64_t n, int32_t max_nb,
                           const jtvae_mess_index_t* mess, jtvae_graph_tables_t* tables, jtvae_graph_csr_t* csr,
                           int32_t n_threads);

/* Junction trees stored back to back (what fast_jtnn/mol_tree.py:78-121 leaves of a MolTree for the encoder:
 * word ids and neighbour lists; the root is a tree's first node). */
typedef struct jtvae_tree_set {
  int64_t n_trees;
  const int64_t* node_ptr; /* [n_trees + 1]                                                  */
  const int32_t* wid;      /* [nodes]                                                        */
  const int64_t* nbr_ptr;  /* [nodes + 1]                                                    */
  const int32_t* nbr;      /* neighbour lists in order: positions inside the tree            */
  const int64_t* cand_ptr; /* [nodes + 1] candidates of every node in a candidate set, or NULL */
  const int32_t* label;    /* [nodes] position of the true candidate among them, or NULL     */
} jtvae_tree_set_t;

typedef struct jtvae_tree_tables { /* (fnode, fmess, node_graph, mess_graph, scope) of jtnn_enc.py:88-92 */
  int64_t* fnode;      /* [nodes]                   */
  int64_t* fmess;      /* [messages]                */
  int64_t* node_graph; /* [nodes, Kn]               */
  int64_t* mess_graph; /* [messages, Km]            */
  int64_t* scope;      /* [n, 2]                    */
  int32_t Kn, Km;      /* in: the widths jtvae_tensorize_tree_counts reported */
} jtvae_tree_tables_t;

typedef struct jtvae_tree_csr {
  int32_t* wid;      /* [nodes]                                                              */
  int32_t* src;      /* [messages] node a message leaves (fmess)                             */
  int32_t* dst;      /* [messages] node it arrives at                                        */
  int32_t* n_rowptr; /* [nodes + 1]    messages arriving at every node (node_graph)          */
  int32_t* n_col;    /* [messages - 1]                                                       */
  int32_t* m_rowptr; /* [messages + 1] messages every message is computed from (mess_graph)  */
  int32_t* m_col;    /* capacity m_cap                                                       */
  int64_t m_cap;
  int32_t* sf;       /* [messages] or NULL: iteration from which a message keeps its value   */
                     /* (see jtvae_prep_stable_from, uncapped: 1 << 30 on a cycle)           */
  int32_t* roots;    /* [n] or NULL: first node of every tree                                */
  int64_t nnz;       /* out: entries of m                                                    */
  int32_t max_sf;    /* out: largest sf (0 without messages)                                 */
} jtvae_tree_csr_t;

/* counts[5] = {nodes, messages incl. the padding message, Kn, Km, m_cap}.  want_widths == 0 (enough for the csr
 * form): Kn, Km come back 0 and m_cap is the cheaper bound sum over nodes of arriving x leaving messages. */
int jtvae_tensorize_tree_counts(const jtvae_tree_set_t* set, const int64_t* ids, int64_t n, int32_t want_widths,
                                int32_t n_threads, int64_t* counts);

/* JTNNEncoder.tensorize after set_batch_nodeID (fast_jtnn/jtnn_enc.py:44-92, datautils.py:166-172): nodes are
 * numbered tree by tree in selection order, messages node by node in neighbour order from 1. */
int jtvae_tensorize_trees(const jtvae_tree_set_t* set, const int64_t* ids, int64_t n, jtvae_tree_tables_t* tables,
                          jtvae_tree_csr_t* csr, int32_t n_threads);

/* The candidate list of a training batch (fast_jtnn/datautils.py:150-161) and the segments of the assembly
 * loss (fast_jtnn/jtnn_vae.py:116-135), from the cand_ptr / label arrays of the tree set: nodes that are leaves
 * (one neighbour) or have exactly one candidate are skipped.  counts[2] = {candidates, segments}.  Out (second
 * call): cand_ids [candidates] graph ids in the candidate set, batch_idx [candidates] tree position in the
 * selection, node_base [candidates] (see jtvae_mess_index_t), seg_len / label_pos [segments] (label_pos is the
 * index of the true candidate in the candidate list; JTVAE_E_RANGE if a segment has none). */
int jtvae_select_candidates_counts(const jtvae_tree_set_t* set, const int64_t* ids, int64_t n, int64_t* counts);
int jtvae_select_candidates(const jtvae_tree_set_t* set, const int64_t* ids, int64_t n, int64_t* cand_ids,
                            int32_t* batch_idx, int64_t* node_base, int32_t* seg_len, int32_t* label_pos);

#ifdef __cplusplus
}
#endif
#endif /* JTVAE_HIP_H */
