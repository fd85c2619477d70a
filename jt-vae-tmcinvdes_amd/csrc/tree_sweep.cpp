// Level sweeps of the compute-once tree GRU as ONE host call each.  A level is three (forward) or four
// (backward) dependent launches of a few microseconds; issued one by one from an interpreted host language the
// per-call overhead of ~80 calls is a millisecond per step, which is what bounds small batches and batches whose
// index tables are prepared concurrently.  The sweeps below issue exactly the launches, in exactly the order, that
// a caller would issue level by level through the per-launch entry points - results are bit-identical - and
// return at the first error.
#include <cstdint>

#include "../../include/jtvae_hip.h"

extern "C" {

int jtvae_tree_gru_forward(const jtvae_tree_fwd_t* a, void* stream) {
  if (!a) return JTVAE_E_NULL;
  if (a->n_levels < 0 || a->H <= 0 || a->ld < a->H) return JTVAE_E_SHAPE;
  if (a->n_levels > 0 && (!a->f_rows || !a->f_n || !a->u_rows || !a->u_n)) return JTVAE_E_NULL;
  for (int32_t L = 0; L < a->n_levels; ++L) {
    if (a->f_n[L] == 0) break;  // every message has settled
    // Uh = U_r h + b for the messages that settled in the previous level and are read by someone
    jtvae_epilogue_t e = jtvae_epilogue_t();
    e.zero_row = -1;
    e.bias = a->bu;
    e.a_rows = a->u_rows[L];
    e.c_rows = a->u_rows[L];
    int rc = jtvae_gemm_nt(a->h, a->ld, a->Ur, a->ld, a->uh, a->ld, a->u_n[L], a->H, a->H, &e, stream);
    if (rc) return rc;
    rc = jtvae_gru_gate_gather(a->h, a->uh, 0, L + 1, a->sf, a->Pr, a->wid, a->rowptr, a->col, a->f_rows[L],
                               a->f_n[L], a->s, a->g, a->t, a->H, a->ld, stream);
    if (rc) return rc;
    rc = jtvae_gru_update(a->s, a->g, a->ld, a->Wzh, a->Whh, a->ld, a->Pz, a->Ph, a->ld, a->wid, a->f_rows[L],
                          a->f_n[L], a->z, a->p, a->h, a->H, a->pad_row, stream);
    if (rc) return rc;
  }
  return 0;
}

int jtvae_tree_gru_backward(const jtvae_tree_bwd_t* a, void* stream) {
  if (!a) return JTVAE_E_NULL;
  if (a->n_levels < 0 || a->H <= 0 || a->ld < a->H) return JTVAE_E_SHAPE;
  if (a->n_levels > 0 && (!a->e_rows || !a->e_n || !a->n_dep || !a->off_e || !a->off_j)) return JTVAE_E_NULL;
  for (int32_t L = 0; L < a->n_levels; ++L) {
    const int32_t* rows = a->e_rows[L];
    const int32_t n = a->e_n[L];
    int rc;
    if (L > 0) {
      float* du = a->du + (size_t)a->off_j[L] * a->ld;
      // dUh, and dh = (gradient from outside) + direct part, for rows whose readers have all been visited
      rc = jtvae_gru_bwd_gather(a->h, a->uh, 0, 0, a->sf, a->active, a->Pr, a->wid, a->dg, a->ds, a->trowptr,
                                a->tcol, rows, n, a->dh, du, a->dhdir, a->H, a->ld, stream);
      if (rc) return rc;
    }
    float* dz = a->dz + (size_t)a->off_e[L] * a->ld;
    float* dp = a->dp + (size_t)a->off_e[L] * a->ld;
    if (L > 0) {
      // dh of the level's rows, and the pre-activation gradients that follow from it
      float* du = a->du + (size_t)a->off_j[L] * a->ld;
      rc = jtvae_gru_bwd_dh(du, a->UrT, a->ld, a->dhdir, a->dh, rows, n, a->z, a->p, a->s, 0, a->T, a->sf, dz, dp,
                            a->ds, a->H, a->ld, a->pad_row, stream);
    } else {
      rc = jtvae_gru_bwd_pointwise(a->dh, a->z, a->p, a->s, 0, a->T, a->sf, rows, n, dz, dp, a->ds, a->H, a->ld,
                                   a->pad_row, stream);
    }
    if (rc) return rc;
    const int32_t nd = a->n_dep[L];  // leading rows of the level that read other messages
    if (nd > 0) {
      jtvae_epilogue_t e0 = jtvae_epilogue_t(), e1 = jtvae_epilogue_t();
      e0.zero_row = e1.zero_row = -1;
      e0.add = a->ds;  // ds += dzpre W_z[:, H:]
      e0.ld_add = a->ld;
      e0.c_rows = rows;
      e1.mode = JTVAE_EPI_MUL_ACC;  // dg = dppre W_h[:, H:];  A_r += dg * t
      e1.aux0 = a->t;
      e1.out2 = a->a_r;
      e1.ld_aux = a->ld;
      e1.c_rows = rows;
      e1.aux_sf = a->sf;
      e1.aux_t = a->T;
      e1.aux_slab_stride = 0;
      jtvae_gemm_nt_problem_t p0 = {dz, a->ld, a->WzhT, a->ld, a->ds, a->ld, nd, a->H, a->H, &e0};
      jtvae_gemm_nt_problem_t p1 = {dp, a->ld, a->WhhT, a->ld, a->dg, a->ld, nd, a->H, a->H, &e1};
      rc = jtvae_gemm_nt_pair(&p0, &p1, stream);
      if (rc) return rc;
    }
  }
  return 0;
}

}  // extern "C"
