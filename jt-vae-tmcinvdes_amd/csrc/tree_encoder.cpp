// JTNNEncoder's forward and backward pass around their level sweeps as ONE host call each (see
// jtvae_tree_encoder_forward / _backward in the header): ~45 launches that an interpreted host issues in ~0.45 ms
// and these routines in ~0.1 ms, which is what bounds steps of a few hundred molecules and steps whose next batch
// is being prepared on the same interpreter.
#include <cstdint>

#include <hip/hip_runtime.h>

#include "../../include/jtvae_hip.h"

namespace {

// Events for the fork / join of the helper stream: a small ring per host thread and device.  An event is reused
// after RING further records; a join is always enqueued within a handful of records of its fork.
constexpr int RING = 64, MAX_DEV = 16;
struct EventRing {
  hipEvent_t ev[RING];
  int n = 0, next = 0;
};
thread_local EventRing rings[MAX_DEV];

// mark(): an event recorded behind everything `s` holds so far; after(): `to` continues only once it has happened
int mark(hipStream_t s, hipEvent_t* out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= MAX_DEV) return JTVAE_E_RANGE;
  EventRing& r = rings[dev];
  if (r.n < RING) {
    e = hipEventCreateWithFlags(&r.ev[r.n], hipEventDisableTiming);
    if (e != hipSuccess) return (int)e;
    r.next = r.n++;
  }
  *out = r.ev[r.next];
  r.next = (r.next + 1) % RING;
  e = hipEventRecord(*out, s);
  return e == hipSuccess ? 0 : (int)e;
}
int after(hipStream_t to, hipEvent_t ev) {
  const hipError_t e = hipStreamWaitEvent(to, ev, 0);
  return e == hipSuccess ? 0 : (int)e;
}
int order_after(hipStream_t to, hipStream_t from) {
  hipEvent_t ev;
  const int rc = mark(from, &ev);
  return rc ? rc : after(to, ev);
}

int gather(float* out, const float* src, int32_t n_src, const jtvae_csr_t& c, int32_t H, int32_t ld,
           const float* relu_mask, int32_t accumulate, void* stream) {
  return jtvae_csr_gather_sum(out, src, n_src, nullptr, c.rowptr, c.col, c.n_rows, H, ld, relu_mask, accumulate,
                              nullptr, 0, stream);
}

// gradient of one per-word table (keyed sum of src rows), then of the weight block (and bias) that produced it
int table_grad(const jtvae_tree_enc_bwd_t* a, const float* src, int32_t n_src, const jtvae_csr_t& pieces,
               const jtvae_csr_t& key, float* part, float* dP, float* dW, int32_t lddw, float* db, int slot,
               void* stream) {
  int rc = gather(part, src, n_src, pieces, a->H, a->ld, nullptr, 0, stream);
  if (rc) return rc;
  rc = gather(dP, part, pieces.n_rows, key, a->H, a->ld, nullptr, 0, stream);
  if (rc) return rc;
  return jtvae_gemm_tn(dP, a->ld, nullptr, a->Eu, a->ld, nullptr, dW, lddw, db, a->NW, a->H, a->H, 0, a->ws[slot],
                       a->ws_bytes[slot], stream);
}

int run(const jtvae_tree_enc_bwd_t* a, void* stream, void* side) {
  const int32_t H = a->H, ld = a->ld, B = a->B, NM = a->NM, NW = a->NW;
  hipStream_t main_s = (hipStream_t)stream;
  const bool forked = side != nullptr && side != stream;
  hipStream_t side_s = forked ? (hipStream_t)side : main_s;
  void* sd = forked ? side : stream;
  int rc;
  // root stage: ReLU mask of the output layer                                              (jtnn_enc.py:33)
  if (a->dtv) {
    rc = gather(a->dpre_root, a->dtv, B, a->ident, H, ld, a->tv, 0, stream);
    if (rc) return rc;
  } else {
    hipError_t e = hipMemsetAsync(a->dpre_root, 0, (size_t)B * ld * sizeof(float), main_s);
    if (e != hipSuccess) return (int)e;
  }
  // W_out[:, H:] and everything that hangs on the output table: nothing in the sweep waits for them
  if (forked && (rc = order_after(side_s, main_s))) return rc;
  rc = jtvae_gemm_tn(a->dpre_root, ld, nullptr, a->nsum, ld, nullptr, a->dWout + H, 2 * H, nullptr, B, H, H, 0,
                     a->ws[0], a->ws_bytes[0], sd);
  if (rc) return rc;
  rc = table_grad(a, a->dpre_root, B, a->o_pieces, a->o_key, a->part_side, a->dPo, a->dWout, 2 * H, a->dbout, 1, sd);
  if (rc) return rc;
  hipEvent_t root_done = nullptr;
  if (forked && (rc = mark(side_s, &root_done))) return rc;
  // back to the messages that reach a root                                                 (jtnn_enc.py:31-32)
  rc = jtvae_gemm_nt(a->dpre_root, ld, a->WohT, ld, a->dnsum, ld, B, H, H, nullptr, stream);
  if (rc) return rc;
  float* dh = a->sweep->dh;
  if (a->dmess) {
    rc = jtvae_copy_rows(a->dmess, a->ld_dmess, dh, ld, NM, H, stream);
    if (rc) return rc;
  }
  rc = gather(dh, a->dnsum, B, a->root_t, H, ld, nullptr, a->dmess ? 1 : 0, stream);
  if (rc) return rc;
  {
    // the sweep runs beside the graph encoder's backward pass from its first level to its last
    const int before = a->crowded ? jtvae_set_small_tile_rows(0) : -2;
    rc = jtvae_tree_gru_backward(a->sweep, stream);
    if (before != -2) jtvae_set_small_tile_rows(before);
    if (rc) return rc;
  }
  // the root group has long finished: its stream now takes the recurrence's weight gradients, one contraction each
  // over every (iteration, message) pair that carried gradient, operands picked through row lists
  if (forked && (rc = order_after(side_s, main_s))) return rc;
  rc = jtvae_gemm_tn(a->sweep->dz, ld, a->zx, a->ss, ld, a->zy, a->dWz + H, 2 * H, nullptr, a->n_z, H, H, 0, a->ws[2],
                     a->ws_bytes[2], sd);
  if (rc) return rc;
  rc = jtvae_gemm_tn(a->sweep->dp, ld, a->zx, a->gs, ld, a->zy, a->dWh + H, 2 * H, nullptr, a->n_z, H, H, 0, a->ws[3],
                     a->ws_bytes[3], sd);
  if (rc) return rc;
  rc = jtvae_gemm_tn(a->sweep->du, ld, a->ux, a->hs, ld, a->uy, a->dUr, H, a->dbu, a->n_u, H, H, 0, a->ws[4],
                     a->ws_bytes[4], sd);
  if (rc) return rc;
  // table gradients of the update gate, the candidate state and the reset gate
  rc = table_grad(a, a->sweep->dz, a->n_e, a->zh_pieces, a->zh_key, a->part_main, a->dPz, a->dWz, 2 * H, a->dbz, 5,
                  stream);
  if (rc) return rc;
  rc = table_grad(a, a->sweep->dp, a->n_e, a->zh_pieces, a->zh_key, a->part_main, a->dPh, a->dWh, 2 * H, a->dbh, 6,
                  stream);
  if (rc) return rc;
  rc = table_grad(a, a->sweep->a_r, NM, a->r_pieces, a->r_key, a->part_main, a->dPr, a->dWr, H, nullptr, 7, stream);
  if (rc) return rc;
  // embedding gradient: dEu = sum over the tables of dP W[:, :H], scattered to the rows of the words used
  const float* dP[4] = {a->dPz, a->dPh, a->dPo, a->dPr};
  const float* WT[4] = {a->WzxT, a->WhxT, a->WoxT, a->WrT};
  for (int i = 0; i < 4; ++i) {
    if (i == 2 && forked && (rc = after(main_s, root_done))) return rc;  // dPo comes from the helper stream
    jtvae_epilogue_t e = jtvae_epilogue_t();
    e.zero_row = -1;
    if (i > 0) {
      e.add = a->dEu;
      e.ld_add = ld;
    }
    rc = jtvae_gemm_nt(dP[i], ld, WT[i], ld, a->dEu, ld, NW, H, H, &e, stream);
    if (rc) return rc;
  }
  rc = gather(a->dE_p, a->dEu, NW, a->word_t, H, ld, nullptr, 0, stream);
  if (rc) return rc;
  rc = jtvae_copy_rows(a->dE_p, ld, a->dE, H, a->V, H, stream);
  if (rc) return rc;
  // the recurrence's weight gradients are not read here: their stream is joined last, so the embedding chain above
  // runs beside them
  if (forked && (rc = order_after(main_s, side_s))) return rc;
  return 0;
}

int run_forward(const jtvae_tree_enc_fwd_t* a, void* stream, void* side) {
  const int32_t H = a->H, ld = a->ld, NW = a->NW;
  const jtvae_tree_fwd_t* sw = a->sweep;
  hipStream_t main_s = (hipStream_t)stream;
  const bool forked = side != nullptr && side != stream;
  hipStream_t side_s = forked ? (hipStream_t)side : main_s;
  void* sd = forked ? side : stream;
  // embedding rows of the words that occur in the batch                                    (jtnn_enc.py:27-28)
  int rc = gather(a->Eu, a->E, a->V, a->word, H, ld, nullptr, 0, stream);
  if (rc) return rc;
  hipEvent_t r_done = nullptr, o_done = nullptr;
  {
    // With the graph encoder on another stream the table products start beside its first large products: they
    // then take the tiling that finds room on a busy CU.
    const int before = (a->crowded & 1) ? jtvae_set_small_tile_rows(0) : -2;
    // the reset-gate table is first read by level 1 of the sweep and the output table after it: with a helper
    // stream they leave the chain of dependent launches that the sweep is
    if (forked) rc = order_after(side_s, main_s);
    jtvae_epilogue_t e = jtvae_epilogue_t();
    e.zero_row = -1;
    if (!rc) rc = jtvae_gemm_nt(a->Eu, ld, a->Wr, ld, const_cast<float*>(sw->Pr), ld, NW, H, H, &e, sd);
    if (!rc && forked) rc = mark(side_s, &r_done);
    e.bias = a->bout;
    if (!rc) rc = jtvae_gemm_nt(a->Eu, ld, a->Wox, ld, a->Po, ld, NW, H, H, &e, sd);
    if (!rc && forked) rc = mark(side_s, &o_done);
    e.bias = a->bz;
    if (!rc) rc = jtvae_gemm_nt(a->Eu, ld, a->Wzx, ld, const_cast<float*>(sw->Pz), ld, NW, H, H, &e, stream);
    e.bias = a->bh;
    if (!rc) rc = jtvae_gemm_nt(a->Eu, ld, a->Whx, ld, const_cast<float*>(sw->Ph), ld, NW, H, H, &e, stream);
    if (before != -2) jtvae_set_small_tile_rows(before);
    if (rc) return rc;
  }
  if (forked && (rc = after(main_s, r_done))) return rc;
  // iteration 0 (h = 0), then every later level                                            (jtnn_enc.py:107-128)
  rc = jtvae_gru_iter0(sw->Pz, sw->Ph, sw->wid, a->train ? sw->z : nullptr, a->train ? sw->p : nullptr, sw->h, a->NM,
                       H, ld, sw->pad_row, stream);
  if (rc) return rc;
  {
    // in a large batch the graph encoder is still running when the sweep reaches its short levels
    const int before = (a->crowded & 2) ? jtvae_set_small_tile_rows(0) : -2;
    rc = jtvae_tree_gru_forward(sw, stream);
    if (before != -2) jtvae_set_small_tile_rows(before);
  }
  if (rc) return rc;
  if (forked && (rc = after(main_s, o_done))) return rc;
  // output stage on the root nodes only                                                    (jtnn_enc.py:31-41)
  rc = gather(a->nsum, sw->h, a->NM, a->root, H, ld, nullptr, 0, stream);
  if (rc) return rc;
  jtvae_epilogue_t e = jtvae_epilogue_t();
  e.zero_row = -1;
  e.add = a->Po;
  e.ld_add = ld;
  e.add_idx = a->wid_root;
  e.act = JTVAE_ACT_RELU;
  return jtvae_gemm_nt(a->nsum, ld, a->Woh, ld, a->tv, ld, a->B, H, H, &e, stream);
}

}  // namespace

extern "C" int jtvae_tree_encoder_forward(const jtvae_tree_enc_fwd_t* a, void* stream, void* side) {
  if (!a || !a->sweep) return JTVAE_E_NULL;
  if (a->H <= 0 || a->ld < a->H || a->B <= 0 || a->NM <= 0 || a->NW <= 0 || a->V <= 0 || a->sweep->H != a->H ||
      a->sweep->ld != a->ld)
    return JTVAE_E_SHAPE;
  if (!a->E || !a->Wzx || !a->Whx || !a->Wox || !a->Wr || !a->Woh || !a->wid_root || !a->Eu || !a->Po || !a->nsum ||
      !a->tv || !a->sweep->Pz || !a->sweep->Pr || !a->sweep->Ph || !a->sweep->h || !a->sweep->wid ||
      (a->train && (!a->sweep->z || !a->sweep->p)))
    return JTVAE_E_NULL;
  return run_forward(a, stream, side);
}

extern "C" int jtvae_tree_encoder_backward(const jtvae_tree_enc_bwd_t* a, void* stream, void* side) {
  if (!a || !a->sweep) return JTVAE_E_NULL;
  if (a->H <= 0 || a->ld < a->H || a->B <= 0 || a->NM <= 0 || a->NW <= 0 || a->V <= 0 || a->n_z <= 0 || a->n_u <= 0 ||
      a->n_e <= 0)
    return JTVAE_E_SHAPE;
  if (!a->Eu || !a->nsum || !a->tv || !a->WohT || !a->WzxT || !a->WhxT || !a->WoxT || !a->WrT || !a->ss || !a->gs ||
      !a->hs || !a->zx || !a->zy || !a->ux || !a->uy || !a->dpre_root || !a->dnsum || !a->part_side || !a->part_main ||
      !a->dPo || !a->dPz || !a->dPh || !a->dPr || !a->dEu || !a->dE_p || !a->dWout || !a->dWz || !a->dWh || !a->dWr ||
      !a->dUr || !a->dbout || !a->dbz || !a->dbh || !a->dbu || !a->dE)
    return JTVAE_E_NULL;
  for (int i = 0; i < 8; ++i)
    if (!a->ws[i]) return JTVAE_E_NULL;
  return run(a, stream, side);
}
