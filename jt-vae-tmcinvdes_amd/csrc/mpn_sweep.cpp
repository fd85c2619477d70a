// Forward and backward of the graph message passing network (MPN / JTMPN) as ONE host call each.  The pass is a
// fixed sequence of a dozen to a score of launches; issued one by one from an interpreted host language the
// per-call overhead is what bounds small batches.  The routines below issue exactly the launches, in exactly the
// order, that a caller would issue through the per-launch entry points - results are bit-identical - and return
// at the first error.
#include <cstdint>

#include <hip/hip_runtime.h>

#include "../../include/jtvae_hip.h"

namespace {

// out[rows] = sum over the CSR of (tree rows | bond rows): the two-source form when tree messages are present
int gather(float* out, const float* tree, int32_t n_tree, const float* bonds, int32_t n_bonds, const int32_t* rowptr,
           const int32_t* col, int32_t n_rows, int32_t H, int32_t ld, int32_t relu_src, void* stream) {
  if (n_tree > 0)
    return jtvae_csr_gather_sum(out, tree, n_tree, bonds, rowptr, col, n_rows, H, ld, nullptr, 0, nullptr, relu_src,
                                stream);
  return jtvae_csr_gather_sum(out, bonds, n_bonds, nullptr, rowptr, col, n_rows, H, ld, nullptr, 0, nullptr, relu_src,
                              stream);
}

}  // namespace

extern "C" {

int jtvae_graph_mpn_forward(const jtvae_mpn_fwd_t* a, void* stream) {
  if (!a) return JTVAE_E_NULL;
  if (a->depth < 1 || a->H <= 0 || a->ld < a->H || a->n_bonds < 0 || a->n_atoms < 0 || a->n_seg < 0 || a->n_tree < 0)
    return JTVAE_E_SHAPE;
  if (!a->msg || (a->depth > 1 && !a->nei) || !a->anei || !a->ah || !a->out) return JTVAE_E_NULL;
  const int32_t NB = a->n_bonds, NA = a->n_atoms, H = a->H, ld = a->ld;
  // binput = W_i fbonds; message 0 = relu(binput) is formed by its readers                 (mpn.py:87-88)
  int rc = jtvae_gemm_nt(a->fbonds, a->ld_fb, a->Wi, a->ld_wi, a->msg[0], ld, NB, H, a->fb_dim, nullptr, stream);
  if (rc) return rc;
  for (int32_t i = 1; i < a->depth; ++i) {  // mpn.py:90-94
    rc = gather(a->nei[i - 1], a->tree, a->n_tree, a->msg[i - 1], NB, a->b_rowptr, a->b_col, NB, H, ld, i == 1, stream);
    if (rc) return rc;
    jtvae_epilogue_t e = jtvae_epilogue_t();
    e.zero_row = -1;
    e.add = a->msg[0];
    e.ld_add = ld;
    e.act = JTVAE_ACT_RELU;
    rc = jtvae_gemm_nt(a->nei[i - 1], ld, a->Wh, ld, a->msg[i], ld, NB, H, H, &e, stream);
    if (rc) return rc;
  }
  // atom readout                                                                           (mpn.py:96-99)
  rc = gather(a->anei, a->tree, a->n_tree, a->msg[a->depth - 1], NB, a->a_rowptr, a->a_col, NA, H, ld, a->depth == 1,
              stream);
  if (rc) return rc;
  jtvae_epilogue_t e0 = jtvae_epilogue_t();
  e0.zero_row = -1;
  e0.bias = a->bo;
  rc = jtvae_gemm_nt(a->fatoms, a->ld_fa, a->Woa, a->ld_woa, a->ah, ld, NA, H, a->fa_dim, &e0, stream);
  if (rc) return rc;
  jtvae_epilogue_t e1 = jtvae_epilogue_t();
  e1.zero_row = -1;
  e1.add = a->ah;
  e1.ld_add = ld;
  e1.act = JTVAE_ACT_RELU;
  rc = jtvae_gemm_nt(a->anei, ld, a->Woh, ld, a->ah, ld, NA, H, H, &e1, stream);
  if (rc) return rc;
  return jtvae_segment_mean(a->out, a->ah, a->seg, a->n_seg, H, ld, stream);  // mpn.py:102-107
}

int jtvae_graph_mpn_backward(const jtvae_mpn_bwd_t* a, void* stream) {
  if (!a) return JTVAE_E_NULL;
  if (a->depth < 1 || a->H <= 0 || a->ld < a->H || a->n_bonds < 0 || a->n_atoms < 0 || a->n_tree < 0)
    return JTVAE_E_SHAPE;
  if (!a->msg || (a->depth > 1 && !a->nei) || !a->dout || !a->dpre_o || !a->danei || !a->dpre || !a->dbinput ||
      !a->dW_i || !a->dW_h || !a->dW_o || !a->db_o || (a->depth > 1 && !a->dnei) || (a->n_tree > 0 && !a->dtree))
    return JTVAE_E_NULL;
  const int32_t NB = a->n_bonds, NA = a->n_atoms, TM = a->n_tree, H = a->H, ld = a->ld;
  hipStream_t st = (hipStream_t)stream;
  // through the mean readout and the ReLU of W_o                                           (mpn.py:99-105)
  int rc = jtvae_segment_mean_bwd(a->dpre_o, a->dout, a->ah, a->seg, a->seg_of, NA, H, ld, stream);
  if (rc) return rc;
  rc = jtvae_gemm_tn(a->dpre_o, ld, nullptr, a->fatoms, a->ld_fa, nullptr, a->dW_o, a->ld_dwo, nullptr, NA, H,
                     a->fa_dim, 0, a->workspace, a->workspace_bytes, stream);
  if (rc) return rc;
  rc = jtvae_gemm_tn(a->dpre_o, ld, nullptr, a->anei, ld, nullptr, a->dW_o + a->fa_dim, a->ld_dwo, a->db_o, NA, H, H, 0,
                     a->workspace, a->workspace_bytes, stream);
  if (rc) return rc;
  rc = jtvae_gemm_nt(a->dpre_o, ld, a->WohT, ld, a->danei, ld, NA, H, H, nullptr, stream);
  if (rc) return rc;
  if (NB > 0) {
    hipError_t he = hipMemsetAsync(a->dbinput, 0, (size_t)NB * ld * sizeof(float), st);
    if (he != hipSuccess) return (int)he;
  }
  // transposed gather of `src` into (tree rows | bond rows): the bond rows masked by the ReLU of the message they
  // belong to and added into dbinput, the tree rows accumulated over the calls
  auto gather_t = [&](const float* src, int32_t n_src, const int32_t* trowptr, const int32_t* tcol, const float* mask,
                      bool first_tree) -> int {
    int r = jtvae_csr_gather_sum(a->dpre, src, n_src, nullptr, trowptr + TM, tcol, NB, H, ld, mask, 0, a->dbinput, 0,
                                 stream);
    if (r) return r;
    if (TM > 0)
      r = jtvae_csr_gather_sum(a->dtree, src, n_src, nullptr, trowptr, tcol, TM, H, ld, nullptr, first_tree ? 0 : 1,
                               nullptr, 0, stream);
    return r;
  };
  rc = gather_t(a->danei, NA, a->at_rowptr, a->at_col, a->msg[a->depth - 1], true);
  if (rc) return rc;
  for (int32_t i = a->depth - 1; i >= 1; --i) {  // message i = relu(binput + W_h nei_i)
    rc = jtvae_gemm_tn(a->dpre, ld, nullptr, a->nei[i - 1], ld, nullptr, a->dW_h, H, nullptr, NB, H, H,
                       i != a->depth - 1 ? 1 : 0, a->workspace, a->workspace_bytes, stream);
    if (rc) return rc;
    rc = jtvae_gemm_nt(a->dpre, ld, a->WhT, ld, a->dnei, ld, NB, H, H, nullptr, stream);
    if (rc) return rc;
    rc = gather_t(a->dnei, NB, a->bt_rowptr, a->bt_col, a->msg[i - 1], false);
    if (rc) return rc;
  }
  if (a->depth == 1) {
    hipError_t he = hipMemsetAsync(a->dW_h, 0, (size_t)H * H * sizeof(float), st);
    if (he != hipSuccess) return (int)he;
  }
  return jtvae_gemm_tn(a->dbinput, ld, nullptr, a->fbonds, a->ld_fb, nullptr, a->dW_i, a->ld_dwi, nullptr, NB, H,
                       a->fb_dim, 0, a->workspace, a->workspace_bytes, stream);
}

}  // extern "C"
