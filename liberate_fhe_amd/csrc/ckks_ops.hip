// ckks_ops.hip — whole engine ops behind ONE C entry each: lf_cc_mult_evk, lf_switch_key.
//
// The reference issues a cc_mult + relinearize as ~250 Python-level calls into its extension (ckks_engine.py:1072-1151,
// 746-961); this library's engine brought that down to a dozen fused launches, but each still cost a Python -> ctypes
// round trip with twenty-odd marshalled arguments (117 us of host time per silver cc_mult, more than the device time of
// a rank of the limb-sharded path).  Here the launches of an op are enqueued by one native call: everything that does
// not change between calls — constants, twiddle tables, key-switch descriptors, scratch — sits in an `lf_ks_plan` the
// caller fills once per (device, level).  The entries only compose the library's own exported steps (same kernels, same
// results); they apply when every limb of the level lives on this device (no exchange step in the middle).
#include "../../include/ckks_hip.h"
#include "ckks_common.h"

extern int lf_g_intt_digits;   // ckks_ks.hip (lf_tune)

extern "C" {

static int plan_ok(const lf_ks_plan *p) {
    return p && p->logN > NTT_TILE_LOG_MAX && p->logN <= 2 * NTT_TILE_LOG_MAX && p->ell >= 1 && p->K >= 1 && p->K <= KS_MAX_K &&
           p->nparts >= 1 && p->dig_nparts >= 0 && p->max_nct >= 1 && p->ql && p->qh && p->kl && p->kh && p->_2q && p->Rs && p->Ninv && p->q_host &&
           p->psi && p->ipsi && p->psi_dp && p->ipsi_dp && p->dig_desc && p->dig_tab && p->ext_desc && p->E && p->Ed && p->PiR &&
           p->state && p->ext && p->sum && p->md_ws;
}

// cc_mult's operand stack x4 in planes format (include/ckks_hip.h LF_NTT_PLANES): decided per call from lf_tune's knob; the pieces
// of a sharded op are separate calls, and a stack written under one setting is refused by a reader under the other (LF_ERR_STATE)
static int stack_planes(const lf_ks_plan *p) { return lf_stack_planes(p->logN, p->ell, p->q_host); }

static int batch_ok(const lf_ks_plan *p, int nct) {
    return plan_ok(p) && (nct == 1 || nct == 2 || nct == 4) && nct <= p->max_nct;
}

static int moddown_any(const lf_ks_plan *p, const int64_t *const *ss, int64_t *const *outs, const int64_t *const *adds, int count,
                       int64_t gal_pinv, const int64_t *g2q, void *stream) {
    const int64_t N = (int64_t)1 << p->logN;
    // up to two special primes (bronze, silver): the elimination among the special rows is at most one product per
    // coefficient — done inside the mod-down launch (lf_ks_moddown_one), which only READS the level constants
    // lf_ks_moddown_consts wrote behind the pivots of `md_consts` polynomials.  A plan whose workspace was never primed
    // (md_consts = 0) or primed for another count takes the two-launch form, which writes them where its own count puts them.
    if (p->K <= LF_MODDOWN_ONE_MAX_K && count == p->md_consts)
        return lf_ks_moddown_one(ss, outs, adds, count, p->ell, p->K, N, p->md_ws, p->md_ws_words, p->PiR, p->PiP, p->Rs, gal_pinv, g2q,
                                 p->ql, p->qh, p->kl, p->kh, p->device, stream);
    return lf_ks_moddown_ws(ss, outs, adds, count, p->ell, p->K, N, p->md_ws, p->md_ws_words, p->PiR, p->PiP, p->Rs, gal_pinv, g2q, p->ql,
                            p->qh, p->kl, p->kh, p->device, stream);
}

// d2 = x1 * y1 -> inverse transform -> digits of the plan's `nct` operand stacks: in ONE launch behind the tiled pass where a
// digit's limbs fit a column thread (lf_intt_mul_digits: silver, bronze), else the inverse transform and lf_ks_digits(_batch)
static int product_digits(const lf_ks_plan *p, int nct, void *stream) {
    const int ell = p->ell, logN = p->logN, dev = p->device;
    const int64_t N = (int64_t)1 << logN, poly = (int64_t)ell * N;
    const int relaxed_plain = LF_NTT_RELAXED | LF_NTT_PLAIN | (stack_planes(p) ? LF_NTT_PLANES : 0);
    const int64_t xs = nct > 1 ? 4 * poly : poly;   // stride between the operand stacks of a batch
    const int e = !lf_g_intt_digits ? LF_ERR_ARG : lf_intt_mul_digits(p->d2, p->x4 + poly, xs, p->x4 + 3 * poly, xs, nct, ell, logN, p->state, p->dig_nparts, p->K, p->dig_desc,
                                     p->dig_tab, p->ipsi, p->ipsi_dp, p->q_host, p->Ninv, relaxed_plain, p->ql, p->qh, p->kl, p->kh, dev, stream);
    if (e != LF_ERR_ARG) return e;   // launched (0) or a runtime error; LF_ERR_ARG: the shape does not qualify, nothing was launched
    if (int e2 = lf_intt_mul(p->d2, p->x4 + poly, xs, p->x4 + 3 * poly, xs, nct, ell, logN, p->ipsi, p->ipsi_dp, p->q_host, p->Ninv, 2,
                             relaxed_plain, p->ql, p->qh, p->kl, p->kh, dev, stream))
        return e2;
    if (nct == 1)
        return lf_ks_digits(p->d2, p->state, p->dig_nparts, p->dig_desc, p->dig_tab, N, p->ql, p->qh, p->kl, p->kh, dev, stream);
    const int64_t *srcs[4];
    int64_t *states[4];
    for (int t = 0; t < nct; ++t) srcs[t] = p->d2 + t * poly, states[t] = p->state + t * poly;
    return lf_ks_digits_batch(srcs, states, nct, p->dig_nparts, p->dig_desc, p->dig_tab, N, 0, nullptr, p->ql, p->qh, p->kl, p->kh, dev, stream);
}

int lf_cc_mult_evk(const lf_ks_plan *p, const int64_t *const *in, const int64_t *const *row0, const int64_t *ksk,
                   int64_t part_stride, int64_t comp_stride, int64_t row_off, int key_format, int64_t *out0, int64_t *out1,
                   void *stream) {
    if (!plan_ok(p) || !p->rescale_scales || !p->PR || !p->x4 || !p->d2 || !in || !row0 || !ksk || !out0 || !out1) return LF_ERR_ARG;
    const int ell = p->ell, rows = p->ell + p->K, logN = p->logN, dev = p->device;
    const int64_t N = (int64_t)1 << logN, poly = (int64_t)ell * N;
    const int xpl = stack_planes(p);
    const int relaxed_plain = LF_NTT_RELAXED | LF_NTT_PLAIN | (xpl ? LF_NTT_PLANES : 0);
    // x0, x1, y0, y1: both rescales inside the first pass of one batched forward transform (ckks_engine.py:1085-1093)
    if (int e = lf_rescale_ntt(in, row0, 4, p->x4, ell, logN, p->rescale_scales, p->round_at, p->psi, p->psi_dp, p->q_host, p->Rs,
                               relaxed_plain, p->_2q, p->ql, p->qh, p->kl, p->kh, dev, stream))
        return e;
    // d2 = x1 * y1 straight into its inverse transform (1099-1101, 1129), its digits (654-705)
    if (int e = product_digits(p, 1, stream)) return e;
    // key switch of d2 with d0, d1 folded into its sums (654-961, 1117-1151)
    if (int e = lf_relin_core_batch(p->state, 0, 1, p->nparts, rows, logN, p->ext_desc, p->E, p->Ed, ksk, part_stride, comp_stride,
                                    row_off, key_format | (xpl ? LF_STACK_PLANES : 0), p->ext, p->sum, p->psi, p->psi_dp, p->ipsi, p->ipsi_dp, p->Ninv, p->x4, 0, p->PR, ell,
                                    p->own, p->q_host, p->ql, p->qh, p->kl, p->kh, dev, stream))
        return e;
    const int64_t *ss[2] = {p->sum, p->sum + (int64_t)rows * N};
    int64_t *outs[2] = {out0, out1};
    return moddown_any(p, ss, outs, nullptr, 2, 0, nullptr, stream);
}

int lf_switch_key(const lf_ks_plan *p, const int64_t *c0, const int64_t *c1, int64_t gal_pinv, int gal_canonical,
                  const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off, int key_format, int64_t *out0,
                  int64_t *out1, void *stream) {
    if (!plan_ok(p) || !c1 || !ksk || !out0 || !out1) return LF_ERR_ARG;
    const int ell = p->ell, rows = p->ell + p->K, logN = p->logN, dev = p->device;
    const int64_t N = (int64_t)1 << logN;
    const int64_t *g2q = (gal_pinv && gal_canonical) ? p->_2q : nullptr;
    if (int e = lf_ks_digits_galois(c1, p->state, p->dig_nparts, p->dig_desc, p->dig_tab, N, gal_pinv, g2q, p->ql, p->qh, p->kl, p->kh,
                                    dev, stream))
        return e;
    if (int e = lf_ks_core(p->state, p->nparts, rows, logN, p->ext_desc, p->E, p->Ed, ksk, part_stride, comp_stride, row_off, key_format, p->ext,
                           p->sum, p->psi, p->psi_dp, p->ipsi, p->ipsi_dp, p->Ninv, p->q_host, p->ql, p->qh, p->kl, p->kh, dev, stream))
        return e;
    const int64_t *ss[2] = {p->sum, p->sum + (int64_t)rows * N};
    int64_t *outs[2] = {out0, out1};
    const int64_t *adds[2] = {c0, nullptr};
    return moddown_any(p, ss, outs, adds, 2, gal_pinv, g2q, stream);
}

/* ---- batches under one key: nct = 1, 2 or 4 ciphertexts per launch set (plan->max_nct >= nct; scratch of ciphertext t at
 * t times the single-ciphertext size).  The compositions the engine's Python used to issue step by step (_ks_batch,
 * _cc_mult_group) behind one call each. ---- */
int lf_switch_key_batch(const lf_ks_plan *p, int nct, const int64_t *const *c0, const int64_t *const *c1, int64_t gal_pinv,
                        int gal_canonical, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
                        int key_format, int64_t *const *out0, int64_t *const *out1, void *stream) {
    if (!batch_ok(p, nct) || !c1 || !ksk || !out0 || !out1) return LF_ERR_ARG;
    const int ell = p->ell, rows = p->ell + p->K, logN = p->logN, dev = p->device;
    const int64_t N = (int64_t)1 << logN, st_stride = (int64_t)ell * N;
    const int64_t *g2q = (gal_pinv && gal_canonical) ? p->_2q : nullptr;
    int64_t *states[4];
    for (int t = 0; t < nct; ++t) states[t] = p->state + t * st_stride;
    if (int e = lf_ks_digits_batch(c1, states, nct, p->dig_nparts, p->dig_desc, p->dig_tab, N, gal_pinv, g2q, p->ql, p->qh, p->kl, p->kh,
                                   dev, stream))
        return e;
    if (int e = lf_ks_core_batch(p->state, st_stride, nct, p->nparts, rows, logN, p->ext_desc, p->E, p->Ed, ksk, part_stride, comp_stride,
                                 row_off, key_format, p->ext, p->sum, p->psi, p->psi_dp, p->ipsi, p->ipsi_dp, p->Ninv, p->q_host, p->ql,
                                 p->qh, p->kl, p->kh, dev, stream))
        return e;
    const int64_t *ss[8], *adds[8];
    int64_t *outs[8];
    for (int t = 0; t < nct; ++t) {
        ss[2 * t] = p->sum + (int64_t)(2 * t) * rows * N, ss[2 * t + 1] = p->sum + (int64_t)(2 * t + 1) * rows * N;
        outs[2 * t] = out0[t], outs[2 * t + 1] = out1[t];
        adds[2 * t] = c0 ? c0[t] : nullptr, adds[2 * t + 1] = nullptr;
    }
    return moddown_any(p, ss, outs, adds, 2 * nct, gal_pinv, g2q, stream);
}

int lf_cc_mult_evk_batch(const lf_ks_plan *p, int nct, const int64_t *const *in, const int64_t *const *row0, const int64_t *ksk,
                         int64_t part_stride, int64_t comp_stride, int64_t row_off, int key_format, int64_t *const *out0,
                         int64_t *const *out1, void *stream) {
    if (!batch_ok(p, nct) || !p->rescale_scales || !p->PR || !p->x4 || !p->d2 || !in || !row0 || !ksk || !out0 || !out1) return LF_ERR_ARG;
    const int ell = p->ell, rows = p->ell + p->K, logN = p->logN, dev = p->device;
    const int64_t N = (int64_t)1 << logN, poly = (int64_t)ell * N;
    const int xpl = stack_planes(p);
    const int relaxed_plain = LF_NTT_RELAXED | LF_NTT_PLAIN | (xpl ? LF_NTT_PLANES : 0);
    for (int t0 = 0; t0 < nct; t0 += 2) {   // rescale + forward transform of the operands, two pairs (8 polynomials) per launch
        const int n = nct - t0 < 2 ? nct - t0 : 2;
        if (int e = lf_rescale_ntt(in + 4 * t0, row0 + 4 * t0, 4 * n, p->x4 + (int64_t)t0 * 4 * poly, ell, logN, p->rescale_scales,
                                   p->round_at, p->psi, p->psi_dp, p->q_host, p->Rs, relaxed_plain, p->_2q, p->ql, p->qh, p->kl, p->kh, dev,
                                   stream))
            return e;
    }
    // the nct products x1 * y1 through one inverse transform (product on load), their digits
    if (int e = product_digits(p, nct, stream)) return e;
    if (int e = lf_relin_core_batch(p->state, poly, nct, p->nparts, rows, logN, p->ext_desc, p->E, p->Ed, ksk, part_stride, comp_stride,
                                    row_off, key_format | (xpl ? LF_STACK_PLANES : 0), p->ext, p->sum, p->psi, p->psi_dp, p->ipsi, p->ipsi_dp, p->Ninv, p->x4, 4 * poly,
                                    p->PR, ell, p->own, p->q_host, p->ql, p->qh, p->kl, p->kh, dev, stream))
        return e;
    const int64_t *ss[8];
    int64_t *outs[8];
    for (int t = 0; t < nct; ++t) {
        ss[2 * t] = p->sum + (int64_t)(2 * t) * rows * N, ss[2 * t + 1] = p->sum + (int64_t)(2 * t + 1) * rows * N;
        outs[2 * t] = out0[t], outs[2 * t + 1] = out1[t];
    }
    return moddown_any(p, ss, outs, nullptr, 2 * nct, 0, nullptr, stream);
}

/* ---- the halves of an op around the digit exchange of a limb-sharded engine (one process per GPU; ckks_engine.py:746-904:
 * the reference gathers every digit on every GPU before it extends).  The plan describes THIS rank's rows at the level:
 * dig_nparts = the digits it owns, nparts = all digits, state = its own digit rows.
 *   pre    everything up to the digits this rank owns (cc_mult: rescale + forward NTT, x1 * y1, inverse NTT, digits);
 *   fwd    extension + forward NTT of digits first .. first + count - 1 of the gathered buffer (own digits while the
 *          others travel, the foreign runs after the wait);
 *   post   inner product over all digits + inverse NTT + mod-down. ---- */
int lf_cc_mult_evk_pre(const lf_ks_plan *p, const int64_t *const *in, const int64_t *const *row0, int which, void *stream) {
    if (!plan_ok(p) || !p->rescale_scales || !p->x4 || !p->d2 || !(which & 3) || (which & ~3) || ((which & 1) && (!in || !row0)))
        return LF_ERR_ARG;
    const int ell = p->ell, logN = p->logN, dev = p->device;
    const int64_t N = (int64_t)1 << logN, poly = (int64_t)ell * N;
    const int relaxed_plain = LF_NTT_RELAXED | LF_NTT_PLAIN | (stack_planes(p) ? LF_NTT_PLANES : 0);
    // which = 3: everything.  1: only the launch that reads the operands (rescale + column pass) — the one launch of this half
    // whose addresses change from call to call; 2: the rest (fixed addresses of the plan: a caller may replay it from a graph)
    const int64_t *const none[8] = {};   // (lf_rescale_ntt takes up to 8 polynomials; unused by the tiled pass)
    if (which & 1)
        if (int e = lf_rescale_ntt(in, row0, 4, p->x4, ell, logN, p->rescale_scales, p->round_at, p->psi, p->psi_dp, p->q_host, p->Rs,
                                   relaxed_plain | (which == 3 ? 0 : LF_NTT_ONLY_COLS), p->_2q, p->ql, p->qh, p->kl, p->kh, dev, stream))
            return e;
    if (!(which & 2)) return 0;
    if (which == 2)
        if (int e = lf_rescale_ntt(none, none, 4, p->x4, ell, logN, p->rescale_scales, p->round_at, p->psi, p->psi_dp, p->q_host, p->Rs,
                                   relaxed_plain | LF_NTT_ONLY_TILED, p->_2q, p->ql, p->qh, p->kl, p->kh, dev, stream))
            return e;
    return product_digits(p, 1, stream);
}

int lf_switch_key_pre(const lf_ks_plan *p, const int64_t *c1, int64_t gal_pinv, int gal_canonical, void *stream) {
    if (!plan_ok(p) || !c1) return LF_ERR_ARG;
    const int64_t *g2q = (gal_pinv && gal_canonical) ? p->_2q : nullptr;
    return lf_ks_digits_galois(c1, p->state, p->dig_nparts, p->dig_desc, p->dig_tab, (int64_t)1 << p->logN, gal_pinv, g2q, p->ql, p->qh,
                               p->kl, p->kh, p->device, stream);
}

int lf_ks_plan_fwd(const lf_ks_plan *p, const int64_t *digits, int first, int count, int relin, void *stream) {
    if (!plan_ok(p) || !digits || first < 0 || count < 0 || first + count > p->nparts) return LF_ERR_ARG;
    const int rows = p->ell + p->K;
    if (relin)
        return lf_relin_fwd(digits, first, count, rows, p->logN, p->ext_desc, p->E, p->Ed, p->ext, p->psi, p->psi_dp, p->own, p->q_host,
                            p->ql, p->qh, p->kl, p->kh, p->device, stream);
    return lf_ks_fwd(digits, count, rows, p->logN, p->ext_desc + 3 * (int64_t)first, p->E, p->Ed,
                     p->ext + (((int64_t)first * rows) << p->logN), p->psi, p->psi_dp, p->q_host, p->ql, p->qh, p->kl, p->kh, p->device, stream);
}

int lf_cc_mult_evk_post(const lf_ks_plan *p, const int64_t *ksk, int64_t part_stride, int64_t comp_stride, int64_t row_off,
                        int key_format, int64_t *out0, int64_t *out1, int which, void *stream) {
    if (!plan_ok(p) || !p->PR || !p->x4 || !(which & 3) || (which & ~3) || ((which & 1) && !ksk) || ((which & 2) && (!out0 || !out1)))
        return LF_ERR_ARG;
    const int rows = p->ell + p->K;
    const int64_t N = (int64_t)1 << p->logN;
    // which: 1 = inner product + inverse NTT (fixed addresses: plan scratch and the key), 2 = the mod-down that writes out0 / out1
    if (which & 1)
        if (int e = lf_relin_tail(p->nparts, rows, p->logN, ksk, part_stride, comp_stride, row_off,
                                  key_format | (stack_planes(p) ? LF_STACK_PLANES : 0), p->ext, p->sum, p->ipsi,
                                  p->ipsi_dp, p->Ninv, p->x4, p->PR, p->ell, p->own, p->q_host, p->ql, p->qh, p->kl, p->kh, p->device, stream))
            return e;
    if (!(which & 2)) return 0;
    const int64_t *ss[2] = {p->sum, p->sum + (int64_t)rows * N};
    int64_t *outs[2] = {out0, out1};
    return moddown_any(p, ss, outs, nullptr, 2, 0, nullptr, stream);
}

int lf_switch_key_post(const lf_ks_plan *p, const int64_t *c0, int64_t gal_pinv, int gal_canonical, const int64_t *ksk,
                       int64_t part_stride, int64_t comp_stride, int64_t row_off, int key_format, int64_t *out0, int64_t *out1,
                       int which, void *stream) {
    if (!plan_ok(p) || !(which & 3) || (which & ~3) || ((which & 1) && !ksk) || ((which & 2) && (!out0 || !out1))) return LF_ERR_ARG;
    const int rows = p->ell + p->K;
    const int64_t N = (int64_t)1 << p->logN;
    const int64_t *g2q = (gal_pinv && gal_canonical) ? p->_2q : nullptr;
    if (which & 1)   // see lf_cc_mult_evk_post
        if (int e = lf_ks_tail(p->nparts, rows, p->logN, ksk, part_stride, comp_stride, row_off, key_format, p->ext, p->sum, p->ipsi, p->ipsi_dp,
                               p->Ninv, p->q_host, p->ql, p->qh, p->kl, p->kh, p->device, stream))
            return e;
    if (!(which & 2)) return 0;
    const int64_t *ss[2] = {p->sum, p->sum + (int64_t)rows * N};
    int64_t *outs[2] = {out0, out1};
    const int64_t *adds[2] = {c0, nullptr};
    return moddown_any(p, ss, outs, adds, 2, gal_pinv, g2q, stream);
}

}  // extern "C"
