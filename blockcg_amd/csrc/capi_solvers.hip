// include/blockcg_hip.h, part 3 of 3: the solvers.  This file holds the reference's control flow (inc/block_solvers.hpp:91-185,
// SBCGrQ, statement for statement with :NNN tags; BCG :10-45, BCGrQ :50-86; CG / SCG src/standard_solvers.cpp:3-95) and its
// m x m coefficient algebra on the host; every loop over lattice sites is a HIP kernel.
#include "capi_internal.hpp"

namespace bcg_impl {

// Phase A of an iteration: T = (A + sigma0) P ; G = P^dagger T   (:134-140)
int phase_A(bcg_context* c, const bcg_gauge* g, double mass, double sigma0, bcg_field* T, const bcg_field* P, CMat& G) {
  int nb = 0;
  bool folded = false;
  BCG_TRY(apply_shifted(c, g, mass, sigma0, T, P, &nb, &folded));
  if (nb > 0) return finish_gram(c, P->m, nb, G, true, folded);
  return gram(c, P, T, G);
}

// Phase B: Q -= T alpha ; G2 = Q^dagger Q   (:148 and the Gram half of :152)
int rmul(bcg_context* c, bcg_field* y, const bcg_field* x, const CMat& M, double b, bcg::RmulMode mode, const char* name);
// Deferred normalisation of Q (widths with both fused row kernels and room for a second matrix in phase B's LDS: m = 8,
// 16).  The reference stores Q rho^-1 (:152, multiply_upper_triangular_inverse_RHS) and reads it back twice: for the P
// updates (:158, :177) and for the next iteration's Q -= T alpha (:148).  Here phase C forms Q rho^-1 in registers for
// the P updates and does NOT write it; the un-normalised Q stays in memory and the next phase B multiplies it by the same
// rho^-1 (same kernel arithmetic, same order: bit-identical iterates) before subtracting T alpha.  One field pass less per
// iteration: (1 + 4 S) s in phase C instead of (2 + 4 S) s.  BCG_LAZY_Q=0 switches it off.
bool lazy_q_width(const bcg_context* c, int m) {
  return c->lazy_q && fast_rows(c, m) && fast_rmul(c, m) && (m == 8 || m == 16 || (m == 32 && c->lazy_q > 1));
}

// rinv_prev: the stored Q is the previous iteration's un-normalised block, to be multiplied by this first (nullptr: Q as it is)
// Qout (fused kernel only): the new Q is written there and Q keeps the old block (pair_shifts below)
int phase_B(bcg_context* c, bcg_field* Q, const bcg_field* T, const CMat& alpha, CMat& G2, const CMat* rinv_prev = nullptr,
            bcg_field* Qout = nullptr) {
  const int m = Q->m;
  if (!fast_rows(c, m)) {
    if (Qout) BCG_FAIL(c, BCG_ERR_INVALID, "phase B: a separate output needs the fused kernel");
    BCG_TRY(rmul(c, Q, T, -alpha, 0.0, bcg::RMUL_ADD, "block_axpy"));
    return gram(c, Q, Q, G2);
  }
  const CMat na = -alpha;
  const CMat* two[2] = {&na, rinv_prev};
  const double2* Md;
  BCG_TRY(upload_mats(c, m, two, rinv_prev ? 2 : 1, &Md));
  int nb;
  {
    ProfScope ps(c, "phaseB", row_bytes(Q, 3), product_flops(Q, rinv_prev ? 3 : 2));  // [rho^-1,] alpha, Gram
    nb = bcg::launch_phaseB(c->stream, m, rows_of(Q), Q->d, T->d, Md, c->partials, c->row_blocks_B,
                            bcg::GramFold{c->dev_gram, c->fold_tickets},
                            rinv_prev ? Md + static_cast<size_t>(m) * m : nullptr, Qout ? Qout->d : nullptr);
  }
  BCG_TRY(check_launch(c, "phaseB"));
  return finish_gram(c, m, nb, G2, true, /*folded=*/true);
}

// Phase C: Q <- Q rho^{-1} ; X_s += P_s A_s ; P_s <- P_s B_s + Q for the n active shifts
// (:152 second half, :145, :158, :175, :177)
int trisolve(bcg_context* c, bcg_field* y, const CMat& R);
// rinv_out != nullptr (lazy_q_width): Q rho^-1 is used but not stored; *rinv_out = rho^-1 for the next phase B
int phase_C(bcg_context* c, bcg_field* Q, const CMat& rho, bcg_field* const* X, bcg_field* const* P, int n,
            const std::vector<CMat>& A, const std::vector<CMat>& Bm, CMat* rinv_out = nullptr) {
  const int m = Q->m;
  if (!fast_rmul(c, m)) {
    BCG_TRY(trisolve(c, Q, rho));
    for (int s = 0; s < n; ++s) {
      BCG_TRY(rmul(c, X[s], P[s], A[s], 0.0, bcg::RMUL_ADD, "block_axpy"));
      BCG_TRY(rmul(c, P[s], Q, Bm[s], 1.0, bcg::RMUL_XPAY, "block_xpay"));
    }
    return BCG_OK;
  }
  const CMat Rinv = bcg::upper_triangular_inverse(rho);
  if (rinv_out) *rinv_out = Rinv;
  for (int s0 = 0, first = 1, per = 0; first || s0 < n; s0 += per, first = 0) {
    per = bcg::phaseC_max_shifts(m, first != 0);
    const int ns = std::min(per, n - s0);
    std::vector<const CMat*> mats;
    mats.push_back(&Rinv);
    double2* Xp[8];
    double2* Pp[8];
    for (int k = 0; k < ns; ++k) {
      mats.push_back(&A[s0 + k]);
      mats.push_back(&Bm[s0 + k]);
      Xp[k] = X[s0 + k]->d;
      Pp[k] = P[s0 + k]->d;
    }
    const double2* Md;
    BCG_TRY(upload_mats(c, m, mats.data(), static_cast<int>(mats.size()), &Md));
    {
      // the launch that applies rho^-1 reads and writes Q; a later launch of the same iteration (m = 32) re-reads it
      ProfScope ps(c, "phaseC", row_bytes(Q, (first && !rinv_out ? 2 : 1) + 4 * ns),
                   product_flops(Q, (rinv_out || first ? 1 : 0) + 2 * ns));
      bcg::launch_phaseC(c->stream, m, rows_of(Q), Q->d, Xp, Pp, ns, Md, rinv_out ? 2 : first, c->row_blocks_C);
    }
    BCG_TRY(check_launch(c, "phaseC"));
  }
  return BCG_OK;
}

// Several iterations of the shifted systems in one pass (SBCGrQ below; kernels_mfma.hip: k_phaseC_multi).  The reference
// updates X_s and P_s of every active shift in every iteration (:175, :177), but only P_0 is read by the rest of the
// iteration (:135).  So an iteration that is certain to be followed by another one updates shift 0 only and keeps its
// un-normalised residual block: phase B of the next iteration writes the new block into another buffer, and the phase C
// that ends the group (the `depth`-th iteration, or the last one before the loop can stop) applies every deferred
// iteration's updates and its own to the shifts >= 1 with X_s, P_s read and written once.  Same kernel arithmetic on the
// same values in the same order: the fields the caller sees after any number of iterations are bit-identical.  Per group
// of D iterations phase C moves (5 (D-1) + D + 4 S) s instead of D (1 + 4 S) s.
// Memory: D - 2 further fields.  The phase B of the iteration that closes a full group writes the new residual block over
// T, which it reads tile by tile just before (T is dead from there to the next operator application), and one of the
// residual buffers the group releases becomes the next T.  So D = 2 costs no memory at all and is what capacity mode
// runs.  BCG_PAIR_SHIFTS=<depth> (0 or 1: off; default 4, the largest instantiated).  Measured at 64^4,
// m = 16, 4 shifts: 67.0 ms per iteration without, 55.5-56.1 at depth 2, 54.3 at 3, 53.4-53.7 at 4 (profiles/r03_group_depth.txt).
// A single system (n_shifts = 1) has no shifted updates to group, but where X_0's update can wait (x0_may_wait, DeferredX0
// below) a group of three or four iterations still saves field passes: 3 s for each iteration inside it and 9 s for the
// closing one, against 5 s each.  A group of two would save nothing (3 + 7), so there it is depth >= 3 or none.
bool x0_may_wait(const bcg_context* c, int m) {
  return c->defer_x0 && lazy_q_width(c, m) && (m == 8 || m == 16) && !capacity_path(c, m);
}
int pair_shifts_depth(const bcg_context* c, int m, int n_shifts) {
  if (c->pair_shifts < 2 || !fast_rows(c, m) || !fast_rmul(c, m)) return 1;
  if (!lazy_q_width(c, m) && m != 32) return 1;  // m = 8, 16 group the un-normalised blocks; m = 32 the stored ones, in pairs
  int d = std::min(c->pair_shifts, capacity_path(c, m) ? 2 : 4);
  while (d >= 2 && !bcg::phaseC_multi_fits(m, d, n_shifts)) --d;
  if (n_shifts < 2 && (d < 3 || !x0_may_wait(c, m))) return 1;
  return d;
}

// An iteration whose updates of the shifts >= 1 wait for a later phase C
struct DeferredIteration {
  bcg_field* Q = nullptr;        // its un-normalised residual block
  CMat rinv;                     // its rho^-1
  int n_active = 0;              // shifts 1 .. n_active-1 were to be updated (:161)
  std::vector<CMat> A, B;        // their coefficients, by shift
  bool x0_deferred = false;      // its update of X_0 (:145) waits too (DeferredX0 below); then:
  CMat A0, R0;                   //   X_0 += P_0 A0 (A0 = alpha delta_old) was due, and P_0 <- P_0 R0 + q was done (R0 = rho^dagger)
  bool x0_backward = false;      //   ... and its P_0 was NOT kept: the spare-less form of a group of two (DeferredX0)
};

// Deferred update of X_0 (m = 8, 16; BCG_DEFER_X0=0 switches it off; capacity mode: the spare-less form below).  X_0 is never read by the
// iteration (:145 is its only appearance), so like the X_s of the shifted systems it can wait for the pass that closes a
// group of iterations -- but P_0 cannot (the next operator application reads it), and X_0 += P_0 A0 needs the P_0 of ITS
// iteration.  With P_0^(i+1) = P_0^(i) R_i + q_i the group's updates collapse onto the group's FIRST P_0 and the
// normalised residual blocks q_k, which the closing pass holds in registers anyway:
//     sum_{i<n} P_0^(i) A_i  =  P_0^(0) C  +  sum_{k<n-1} q_k D_k ,
//     C = sum_i (R_0 .. R_{i-1}) A_i ,   D_k = sum_{i>k} (R_{k+1} .. R_{i-1}) A_i        (m x m, composed on the host)
// (the closing iteration's own update uses its own P_0 as before).  So an iteration inside a group runs shift 0 as
// Pout = P_0 R + q alone -- three field passes instead of five (k_phaseC_p0) -- and the first one of a group writes into a
// spare field so that P_0^(0) survives until the closing pass, which reads it once more and does n more products.
// The P_0 sequence, hence every coefficient, residual and iteration count, is bit-identical to the undeferred solver; X_0
// differs by rounding (the composed matrices associate the products differently): tolerance-level, tests/test_gpu_parity.py.
// Cost: one more field (the spare).  Measured: DESIGN.md section 4.
// The spare-less form of a group of TWO (capacity mode, whose point is the memory; or a solve that could not allocate the
// spare): the first iteration updates P_0 in place, and the closing pass -- which holds P_0^(1) and q_0 in registers --
// gets the lost block back from P_0^(1) = P_0^(0) R_0 + q_0:
//     P_0^(0) A_0 = (P_0^(1) - q_0) M ,   M = R_0^-1 A_0 = (rho_0^-1)^dagger A_0
// so X_0 += P_0^(1) (A_1 + M) - q_0 M: the closing step's own coefficient changed on the host and ONE more product, no
// field read or kept.  R_0^-1 amplifies rounding by the condition of rho_0 -- the triangular factor of the new residual
// block in the old orthonormal basis, 1.1 .. 1.4 along ordinary solves (it is the block's convergence factor per
// direction) -- so the form is taken only while ||rho||_F ||rho^-1||_F <= 64 m (condition <= 64 m at worst: rounding
// below 1e-12 relative); otherwise that iteration runs the plain five-pass update.  One inverse only: longer groups
// would chain them, and keep the spare.
struct DeferredX0 {
  std::vector<CMat> mats;  // [C, D_0 .. D_{n-2}]
};
DeferredX0 compose_x0(const std::vector<DeferredIteration>& pend) {
  DeferredX0 out;
  const int n = static_cast<int>(pend.size());
  if (n == 0 || !pend[0].x0_deferred || pend[0].x0_backward) return out;
  const int m = pend[0].A0.dim();
  for (int k = -1; k + 1 < n; ++k) {  // k = -1: the coefficient of P_0^(0); k >= 0: that of q_k
    CMat sum(m), chain = CMat::identity(m);
    for (int i = k + 1; i < n; ++i) {
      sum = sum + chain * pend[i].A0;
      chain = chain * pend[i].R0;
    }
    out.mats.push_back(sum);
  }
  return out;
}

// The deferred iterations' updates and the current one's (coefficients A0/B0 for shift 0, Anew/Bnew by shift for the rest).
// rinv_out != nullptr (deferred normalisation, m = 8, 16): the blocks are un-normalised and one launch does everything.
// nullptr (m = 32): the blocks are stored normalised -- the current one by the ordinary phase C launch that also updates
// shift 0.  Either way a launch takes as many shifts as have room for their matrices in LDS (each launch reads the
// residual blocks again, and normalises them again if they are stored un-normalised).
// flush_rinv != nullptr: the "current" iteration is itself a deferred one whose shift 0 has been updated already (error
// paths, sbcgrq_flush_pending): only the shifts >= 1 are touched, *flush_rinv is its rho^-1 and rho_new, A0, B0 are unused.
// p0_first != nullptr: the group's X_0 updates were deferred (DeferredX0); they are added from that field (the group's first
// P_0) and the normalised residual blocks in the launch that takes entry 0
int phase_C_multi(bcg_context* c, const std::vector<DeferredIteration>& pend, bcg_field* Qnew, const CMat& rho_new,
                  bcg_field* const* X, bcg_field* const* P, const CMat& A0, const CMat& B0, int n_active_new,
                  const std::vector<CMat>& Anew, const std::vector<CMat>& Bnew, CMat* rinv_out, bool lazy,
                  const CMat* flush_rinv = nullptr, const bcg_field* p0_first = nullptr) {
  const int m = Qnew->m, ns = static_cast<int>(pend.size()) + 1;
  const CMat rinv_new = flush_rinv ? *flush_rinv : (lazy ? bcg::upper_triangular_inverse(rho_new) : CMat());
  const double2* Qd[4];
  for (int j = 0; j + 1 < ns; ++j) Qd[j] = pend[j].Q->d;
  Qd[ns - 1] = Qnew->d;
  struct Entry {
    int shift, first, last;
    std::vector<const CMat*> mats;
  };
  std::vector<Entry> entries;
  if (flush_rinv) {
    // nothing for shift 0
  } else if (lazy) {
    entries.push_back(Entry{0, ns - 1, ns, {&A0, &B0}});
  } else {
    const std::vector<CMat> a0(1, A0), b0(1, B0);
    BCG_TRY(phase_C(c, Qnew, rho_new, X, P, 1, a0, b0, nullptr));  // Q <- Q rho^-1 stored; shift 0
  }
  const int n_first = ns > 1 ? pend[0].n_active : n_active_new;
  for (int s = 1; s < n_first; ++s) {  // the active set only shrinks: a shift takes a prefix of the steps
    Entry e{s, 0, 0, {}};
    for (int j = 0; j < ns; ++j) {
      const bool on = s < (j + 1 < ns ? pend[j].n_active : n_active_new);
      if (!on) break;
      e.mats.push_back(j + 1 < ns ? &pend[j].A[s] : &Anew[s]);
      e.mats.push_back(j + 1 < ns ? &pend[j].B[s] : &Bnew[s]);
      ++e.last;
    }
    entries.push_back(e);
  }
  DeferredX0 x0 = (p0_first && lazy && !flush_rinv) ? compose_x0(pend) : DeferredX0();
  CMat A0_back;  // the closing step's coefficient in the spare-less form: A_1 + M
  if (!p0_first && lazy && !flush_rinv && ns == 2 && pend[0].x0_deferred && pend[0].x0_backward) {
    const CMat M = pend[0].rinv.adjoint() * pend[0].A0;  // R_0^-1 A_0, R_0 = rho_0^dagger
    A0_back = A0 + M;
    entries[0].mats[0] = &A0_back;
    x0.mats.push_back(CMat(m));  // the slot of the P_0^(0) term: not read (p1 = nullptr)
    x0.mats.push_back(-M);       // q_0's coefficient
  }
  const int xacc = static_cast<int>(x0.mats.size());
  if (xacc > 0) {  // (entry 0 is shift 0's: its step matrices, then the composed ones)
    for (const CMat& M : x0.mats) entries[0].mats.push_back(&M);
  }
  static const char* const names[5] = {"", "", "phaseC_multi2", "phaseC_multi3", "phaseC_multi4"};
  for (size_t e0 = 0; e0 < entries.size();) {
    const bool with_x0 = xacc > 0 && e0 == 0;
    // as many entries as have LDS room for their matrices beside the rinv_j (m = 16: four shifts at any depth, eight at
    // depth 2); every launch reads -- and normalises -- the residual blocks again
    int n = 0;
    for (int room = bcg::phaseC_multi_capacity(m) - (lazy ? ns : 0); e0 + n < entries.size() && n < 8; ++n) {
      room -= static_cast<int>(entries[e0 + n].mats.size());
      if (room < 0) break;
    }
    if (n == 0) BCG_FAIL(c, BCG_ERR_UNSUPPORTED, "phase C: the matrices of one entry do not fit a launch");
    std::vector<const CMat*> mats;
    if (lazy) {
      for (int j = 0; j + 1 < ns; ++j) mats.push_back(&pend[j].rinv);
      mats.push_back(&rinv_new);
    }
    double2* Xp[8];
    double2* Pp[8];
    int first[8], last[8];
    double products = lazy ? ns : 0;
    for (int k = 0; k < n; ++k) {
      const Entry& e = entries[e0 + k];
      Xp[k] = X[e.shift]->d;
      Pp[k] = P[e.shift]->d;
      first[k] = e.first;
      last[k] = e.last;
      mats.insert(mats.end(), e.mats.begin(), e.mats.end());
      products += static_cast<double>(e.mats.size());
    }
    const double2* Md;
    BCG_TRY(upload_mats(c, m, mats.data(), static_cast<int>(mats.size()), &Md));
    {
      // one profile entry per group size: each is its own kernel instantiation (k_phaseC_multi<m, waves, ns>)
      const double2* const p1 = (with_x0 && p0_first) ? p0_first->d : nullptr;
      // (the unread slot of the spare-less form is no product)
      ProfScope ps(c, names[ns], row_bytes(Qnew, ns + 4 * n + (p1 ? 1 : 0)), product_flops(Qnew, products - ((with_x0 && !p1) ? 1 : 0)));
      bcg::launch_phaseC_multi(c->stream, m, rows_of(Qnew), ns, Qd, Xp, Pp, n, first, last, Md, c->row_blocks_C, lazy,
                               with_x0 ? xacc : 0, p1);
    }
    BCG_TRY(check_launch(c, "phaseC_multi"));
    e0 += n;
  }
  if (rinv_out) *rinv_out = rinv_new;
  return BCG_OK;
}

// thinQR (inc/fields.hpp:140-146)
int thin_qr(bcg_context* c, bcg_field* y, CMat& R) {
  CMat G;
  BCG_TRY(gram(c, y, y, G));
  if (!G.all_finite()) BCG_FAIL(c, BCG_ERR_NUMERIC, "thinQR: Gram matrix is not finite");
  if (!bcg::cholesky_upper(G, R)) BCG_FAIL(c, BCG_ERR_NUMERIC, "thinQR: Gram matrix is not positive definite");
  return trisolve(c, y, R);
}

double max_ratio(const std::vector<double>& num, const std::vector<double>& den) {
  double r = 0.0;
  for (size_t i = 0; i < num.size(); ++i) r = std::max(r, num[i] / den[i]);
  return r;
}

}  // namespace bcg_impl

using namespace bcg_impl;

// ---- the callers either side of the hot path, on the same kernels ------------
namespace {

// Re(a^dagger b) for N_rhs = 1 fields: real_dot (inc/fields.hpp:93-99)
int real_dot(bcg_context* c, const bcg_field* a, const bcg_field* b, double& out) {
  CMat G;
  BCG_TRY(gram(c, a, b, G, false));
  out = G(0, 0).real();
  return BCG_OK;
}

struct FieldPool {  // work fields of one solver call, released together
  bcg_context* c;
  std::vector<bcg_field*> f;
  int parity = -1;  // of the fields made without an `init` to copy: set it to the parity of the solve's source
  explicit FieldPool(bcg_context* ctx) : c(ctx) {}
  ~FieldPool() {
    for (bcg_field* p : f) bcg_field_destroy(p);
  }
  int make(int m, bcg_field** out, const bcg_field* init = nullptr) {
    bcg_field* p = nullptr;
    const int par = init ? init->parity : parity;
    if (par >= 0) BCG_TRY(bcg_field_create_half(c, m, par, &p));
    else BCG_TRY(bcg_field_create(c, m, &p));
    f.push_back(p);
    if (init) BCG_TRY(bcg_field_copy(p, init));
    *out = p;
    return BCG_OK;
  }
};

}  // namespace

extern "C" {

// True relative residuals exactly as the reference's tests and benchmark measure them
// (test/solvers.cpp:104-116, benchmark.cpp:93-103): AX = op(X_s) + sigma_s X_s - B ;
// res[s][i] = sqrt( (AX^dagger AX)_ii / (B^dagger B)_ii ).
int bcg_true_residuals(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* const* X, const bcg_field* B,
                       int n_shifts, const double* sigma, double* res_out) {
  DeviceScope on_device(c);
  if (!c || !g || !X || !B || !sigma || !res_out || n_shifts < 1 || g->ctx != c || B->ctx != c) return BCG_ERR_INVALID;
  const int m = B->m;
  FieldPool pool(c);
  pool.parity = B->parity;
  bcg_field* AX = nullptr;  // only the unfused path needs it
  CMat b2, r2;
  BCG_TRY(gram(c, B, B, b2));
  for (int s = 0; s < n_shifts; ++s) {
    if (!same_shape(X[s], B)) return BCG_ERR_INVALID;
    // One pass where the bundle stencil applies (m = 16, whole-field tmp): tmp = D X_s, then the second stencil
    // forms (mass^2 + sigma_s) X_s - D tmp - B in registers and accumulates its Gram product; AX is never written
    // (5 field passes + 2 link passes instead of 9 + 2).
    if (m == 16 && B->parity < 0 && fast_hop(c, m) && !capacity_path(c, m) &&
        bcg::hop_uses_bundle(m, c->lat, kFastBlocks, c->hop_tune, 0, bcg::HopWindow())) {
      bcg_field* tmp;
      BCG_TRY(get_tmp(c, m, &tmp));
      BCG_TRY(hop(c, g, tmp, X[s], bcg::HOP_PLAIN, nullptr, 0.0));
      BCG_TRY(halo_field(c, tmp));
      BCG_TRY(ensure_scratch(c));
      int nb;
      {
        ProfScope ps(c, "hop_residual", alg_bytes(c, m, 3, 1));  // reads tmp, X_s, B and the links; writes nothing
        nb = bcg::launch_hop_fast(c->stream, m, c->lat, g->U, g->Ughost, tmp->d, c->halo_recv, const_cast<double2*>(B->d),
                                  bcg::HOP_RESID, X[s]->d, mass * mass + sigma[s], c->partials, true, kFastBlocks, c->hop_tune, 0);
      }
      if (nb > 0) {
        BCG_TRY(check_launch(c, "hop_residual"));
        BCG_TRY(finish_gram(c, m, nb, r2, true));
        for (int i = 0; i < m; ++i) res_out[s * m + i] = std::sqrt(r2(i, i).real() / b2(i, i).real());
        continue;
      }
    }
    if (!AX) BCG_TRY(pool.make(m, &AX));
    BCG_TRY(apply_shifted(c, g, mass, sigma[s], AX, X[s]));  // op + add(X_s, sigma_s) in one pass
    BCG_TRY(axpby(c, AX, 1.0, B, -1.0, "axpby"));
    BCG_TRY(gram(c, AX, AX, r2));
    for (int i = 0; i < m; ++i) res_out[s * m + i] = std::sqrt(r2(i, i).real() / b2(i, i).real());
  }
  return BCG_OK;
}

// CG (src/standard_solvers.cpp:3-32): single right-hand side, scalar coefficients.
int bcg_cg_solve(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* x, const bcg_field* b, double eps,
                 int max_iterations, int* iterations_out) {
  DeviceScope on_device(c);
  if (!c || !g || !same_shape(x, b) || x == b || g->ctx != c || b->ctx != c) return BCG_ERR_INVALID;
  if (b->m != 1) BCG_FAIL(c, BCG_ERR_INVALID, "CG takes fermion_field arguments (N_rhs = 1)");
  FieldPool pool(c);
  bcg_field *t, *p, *r;
  BCG_TRY(bcg_field_set_zero(x));  // :5
  pool.parity = b->parity;
  BCG_TRY(pool.make(1, &t));
  BCG_TRY(pool.make(1, &p, b));    // :7
  BCG_TRY(pool.make(1, &r, b));    // :8
  double rr, pt;
  BCG_TRY(real_dot(c, r, r, rr));  // :9
  int iter = 0;
  const double stop = eps * std::sqrt(rr);  // :11
  while (std::sqrt(rr) > stop && iter < max_iterations) {  // :13
    BCG_TRY(apply_shifted(c, g, mass, 0.0, t, p));          // :15
    ++iter;
    BCG_TRY(real_dot(c, p, t, pt));
    const double alpha = rr / pt;                           // :18
    BCG_TRY(axpby(c, r, 1.0, t, -alpha, "axpby"));          // :20
    const double rr_old = rr;
    BCG_TRY(real_dot(c, r, r, rr));                         // :23
    const double beta = rr / rr_old;                        // :24
    BCG_TRY(axpby(c, x, 1.0, p, alpha, "axpby"));           // :26
    BCG_TRY(axpby(c, p, beta, r, 1.0, "axpby"));            // :28
  }
  BCG_TRY(stream_sync(c));
  if (iterations_out) *iterations_out = iter;
  return BCG_OK;
}

// SCG (src/standard_solvers.cpp:34-95): multi-shift CG, scalar zeta/theta recurrences.
int bcg_scg_solve(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* const* x, const bcg_field* b, int n_shifts,
                  const double* sigma, double eps, double eps_shifts, int max_iterations, int* iterations_out) {
  DeviceScope on_device(c);
  if (!c || !g || !x || !b || !sigma || n_shifts < 1 || g->ctx != c || b->ctx != c) return BCG_ERR_INVALID;
  if (b->m != 1) BCG_FAIL(c, BCG_ERR_INVALID, "SCG takes fermion_field arguments (N_rhs = 1)");
  for (int s = 0; s < n_shifts; ++s)
    if (!same_shape(x[s], b) || x[s] == b) return BCG_ERR_INVALID;
  if (sigma[0] < 0.0) BCG_FAIL(c, BCG_ERR_INVALID, "SCG: shifts must be zero or positive");              // :40
  if (!std::is_sorted(sigma, sigma + n_shifts)) BCG_FAIL(c, BCG_ERR_INVALID, "SCG: shifts must be ascending");  // :41-42
  int active = n_shifts;                       // :45
  double alpha = 1.0, beta = 0.0;              // :46-47
  std::vector<double> zeta(n_shifts, 1.0), theta(n_shifts, 1.0);  // :48-49
  FieldPool pool(c);
  std::vector<bcg_field*> p(n_shifts);
  for (int s = 0; s < n_shifts; ++s) {
    BCG_TRY(bcg_field_set_zero(x[s]));         // :50-52
    BCG_TRY(pool.make(1, &p[s], b));           // :53
  }
  bcg_field *t, *r;
  pool.parity = b->parity;
  BCG_TRY(pool.make(1, &t));
  BCG_TRY(pool.make(1, &r, b));                // :54
  double rr, pt;
  BCG_TRY(real_dot(c, r, r, rr));              // :55
  int iter = 0;
  const double stop = eps * std::sqrt(rr);     // :57
  while (std::sqrt(rr) > stop && iter < max_iterations) {  // :58
    BCG_TRY(apply_shifted(c, g, mass, sigma[0], t, p[0]));  // :60-61
    ++iter;
    const double alpha_old = alpha;
    BCG_TRY(real_dot(c, p[0], t, pt));
    alpha = rr / pt;                                        // :65
    BCG_TRY(axpby(c, r, 1.0, t, -alpha, "axpby"));          // :67
    const double rr_old = rr;
    BCG_TRY(real_dot(c, r, r, rr));                         // :69
    const double beta_old = beta;
    beta = rr / rr_old;                                     // :71
    // :73-87 -- the updates of all active shifts as ONE pass over r, x_s, p_s (k_scg_update; same expressions as the
    // axpys, so the same iterates): coefficients first, then one launch
    std::vector<double> a_s(active), b_s(active), z_s(active);
    std::vector<double2*> xs(active), ps(active);
    a_s[0] = alpha; b_s[0] = beta; z_s[0] = 1.0;            // :73, :75
    for (int s = active - 1; s > 0; --s) {                  // :76
      double inv_theta = 1.0 + (sigma[s] - sigma[0]) * alpha;              // :78
      inv_theta += beta_old * (alpha / alpha_old) * (1.0 - theta[s]);      // :79
      theta[s] = 1.0 / inv_theta;                                          // :80
      zeta[s] *= theta[s];                                                 // :81
      a_s[s] = alpha * theta[s];                                           // :82  x_s += alpha_s p_s  (:85)
      b_s[s] = beta * theta[s] * theta[s];                                 // :83  p_s = beta_s p_s + zeta_s r  (:87)
      z_s[s] = zeta[s];
    }
    for (int s = 0; s < active; ++s) { xs[s] = x[s]->d; ps[s] = p[s]->d; }
    {
      ProfScope ps_(c, "scg_update");
      bcg::launch_scg_update(c->stream, r->d, active, xs.data(), ps.data(), a_s.data(), b_s.data(), z_s.data(), rows_of(r));
    }
    BCG_TRY(check_launch(c, "scg_update"));
    if (std::sqrt(rr) * zeta[active - 1] < eps_shifts) --active;           // :90-92
  }
  BCG_TRY(stream_sync(c));
  if (iterations_out) *iterations_out = iter;
  return BCG_OK;
}

// BCG (inc/block_solvers.hpp:10-45): block CG without the QR stabilisation.
int bcg_bcg_solve(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* X, const bcg_field* B, double eps,
                  int max_iterations, int* iterations_out) {
  DeviceScope on_device(c);
  if (!c || !g || !same_shape(X, B) || X == B || g->ctx != c || B->ctx != c) return BCG_ERR_INVALID;
  const int m = B->m;
  FieldPool pool(c);
  bcg_field *T, *P, *R;
  BCG_TRY(bcg_field_set_zero(X));   // :14
  pool.parity = B->parity;
  BCG_TRY(pool.make(m, &T));
  BCG_TRY(pool.make(m, &P, B));     // :16
  BCG_TRY(pool.make(m, &R, B));
  CMat r2, r2_old, pt;
  BCG_TRY(gram(c, R, R, r2));       // :17
  std::vector<double> norm0(m);
  for (int i = 0; i < m; ++i) norm0[i] = std::sqrt(r2(i, i).real());  // :19-20
  double residual = 1.0;
  int iter = 0;
  while (residual > eps && iter < max_iterations) {          // :25
    BCG_TRY(apply_shifted(c, g, mass, 0.0, T, P));            // :27
    ++iter;
    BCG_TRY(gram(c, P, T, pt));
    if (!pt.all_finite()) BCG_FAIL(c, BCG_ERR_NUMERIC, "BCG: P^dagger A P is not finite");
    const CMat alpha = bcg::inverse_full_pivot(pt) * r2;      // :31  (P.T)^-1 (R.R)
    BCG_TRY(rmul(c, R, T, -alpha, 0.0, bcg::RMUL_ADD, "block_axpy"));  // :33
    r2_old = r2;
    BCG_TRY(gram(c, R, R, r2));                               // :35
    const CMat beta = bcg::inverse_full_pivot(r2_old) * r2;   // :36
    BCG_TRY(rmul(c, X, P, alpha, 0.0, bcg::RMUL_ADD, "block_axpy"));   // :38
    BCG_TRY(rmul(c, P, R, beta, 1.0, bcg::RMUL_XPAY, "block_xpay"));   // :40
    residual = 0.0;                                           // :41-42
    for (int i = 0; i < m; ++i) residual = std::max(residual, std::sqrt(r2(i, i).real()) / norm0[i]);
  }
  BCG_TRY(stream_sync(c));
  if (iterations_out) *iterations_out = iter;
  return BCG_OK;
}

// BCGrQ (inc/block_solvers.hpp:50-86) is SBCGrQ with the single shift sigma = 0: same statements in the same
// order (X += P alpha delta_old, Q -= T alpha, thinQR, P = P rho^dagger + Q, delta = rho delta).
int bcg_bcgrq_solve(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* X, const bcg_field* B, double eps,
                    int max_iterations, int* iterations_out) {
  DeviceScope on_device(c);
  const double zero = 0.0;
  bcg_field* Xs[1] = {X};
  return bcg_sbcgrq_solve(c, g, mass, Xs, const_cast<bcg_field*>(B), 1, &zero, eps, 0.0, max_iterations, 0, iterations_out,
                          nullptr, nullptr);
}

double bcg_sbcgrq_bytes_per_iteration(const bcg_context* c, int m, int n_shifts) {
  if (!c) return 0.0;
  const double s = 48.0 * m, gl = 144.0 * c->ndim;
  return static_cast<double>(c->lat.V) * ((14.0 + 4.0 * (n_shifts - 1)) * s + 2.0 * gl);
}

// ---- SBCGrQ (inc/block_solvers.hpp:91-185) -----------------------------------------------------
// The solver is a resumable state machine so that a caller can run (and time) an exact number of
// iterations: begin = everything before the while loop (:97-131), iterate = loop bodies, end =
// release of the work fields.  bcg_sbcgrq_solve is begin + iterate(max_iterations) + end.
}  // extern "C"

struct bcg_sbcgrq_state {
  bcg_context* c = nullptr;
  const bcg_gauge* g = nullptr;
  double mass = 0.0;
  int m = 0, n_shifts = 0;
  std::vector<double> sigma;
  double eps = 0.0, eps_shifts = 0.0;
  std::vector<bcg_field*> X;
  bcg_field* B = nullptr;
  bcg_field* T = nullptr;
  bcg_field* Q = nullptr;
  std::vector<bcg_field*> P;
  int n_unconverged = 0;
  CMat alpha, rho, delta, alpha_inv, alpha_inv_old, rho_old;
  std::vector<CMat> alpha_s, beta_s;
  std::vector<double> b_norm;
  double residual = 1.0;
  int iter = 0;
  CMat q_rinv;          // deferred normalisation (lazy_q_width): the stored Q times this is the reference's Q
  bool q_lazy = false;
  // pair_shifts_depth > 1: iterations whose updates of the shifts >= 1 wait for a later phase C, oldest first
  std::vector<DeferredIteration> pending;
  std::vector<bcg_field*> Qfree;         // residual buffers not in use (depth - 2 of them when nothing is pending)
  int depth = 1;
  bool defer_x0 = false;                 // the updates of X_0 wait for the pass that closes the group too (DeferredX0)
  bcg_field* P0_spare = nullptr;         // ... the field the first iteration of a group writes its new P_0 into
  bcg_field* P0_first = nullptr;         // ... and, inside a group, the group's first P_0 (P[0] is then the former spare)
  bool x0_backward = false;              // groups of two without a spare field: the spare-less form (DeferredX0)
  bool failed = false;                   // an iteration returned an error: no further iterations on this state
};

namespace {

void sbcgrq_release(bcg_sbcgrq_state* st) {
  if (st->T && st->T != st->B) bcg_field_destroy(st->T);  // T, Q and the spare buffers rotate: any of them may be B's storage
  if (st->Q && st->Q != st->B) bcg_field_destroy(st->Q);
  for (bcg_field* q : st->Qfree)
    if (q != st->B) bcg_field_destroy(q);
  for (const DeferredIteration& d : st->pending)
    if (d.Q != st->B && d.Q != st->Q) bcg_field_destroy(d.Q);
  st->Qfree.clear();
  st->pending.clear();
  if (st->P0_spare) bcg_field_destroy(st->P0_spare);
  if (st->P0_first) bcg_field_destroy(st->P0_first);
  st->P0_spare = st->P0_first = nullptr;
  for (bcg_field* p : st->P)
    if (p) bcg_field_destroy(p);
  st->T = st->Q = nullptr;
  st->P.clear();
}

// One pass of the loop body, inc/block_solvers.hpp:133-182.
// more_follow: the caller will run at least one more iteration if this one leaves the residual above eps
int sbcgrq_iteration(bcg_sbcgrq_state* st, bcg_sbcgrq_trace* trace, bool more_follow) {
  bcg_context* c = st->c;
  const int m = st->m, n_shifts = st->n_shifts;
  const std::vector<double>& sigma = st->sigma;
  const CMat Identity = CMat::identity(m);
  // T = (A + sigma_0) P_0 ; alpha_inv = P_0^dagger T                          :134-140
  ++st->iter;                                            // :137
  st->alpha_inv_old = st->alpha_inv;                     // :139
  BCG_TRY(phase_A(c, st->g, st->mass, sigma[0], st->T, st->P[0], st->alpha_inv));  // global reduction #1
  if (!st->alpha_inv.all_finite()) BCG_FAIL(c, BCG_ERR_NUMERIC, "SBCGrQ: P^dagger A P is not finite");
  st->alpha = bcg::inverse_full_pivot(st->alpha_inv);    // :142
  const CMat alpha_delta = st->alpha * st->delta;        // :145 uses delta of the previous iteration
  // Q -= T alpha ; Gram matrix of the new Q                                  :148, :152
  CMat G2;
  if (!st->pending.empty()) {  // the old block is needed by a later phase C: the new one goes to another buffer
    // ... a spare one, or, in the iteration that closes a full group, T itself: phase B reads each tile of T just before
    // it writes the same tile of the new Q, and T is not read again before the next operator application rewrites it
    const bool over_T = st->Qfree.empty();
    bcg_field* out = over_T ? st->T : st->Qfree.back();
    BCG_TRY(phase_B(c, st->Q, st->T, st->alpha, G2, st->q_lazy ? &st->q_rinv : nullptr, out));  // global reduction #2
    if (over_T) st->T = nullptr;  // one of the group's buffers takes its place below
    else st->Qfree.pop_back();
    st->Q = out;  // the old buffer stays with pending.back()
  } else {
    BCG_TRY(phase_B(c, st->Q, st->T, st->alpha, G2, st->q_lazy ? &st->q_rinv : nullptr));  // global reduction #2
  }
  st->rho_old = st->rho;                                 // :150
  if (c->debug_fail_iter > 0 && st->iter == c->debug_fail_iter) G2(0, 0) = cd(std::nan(""), 0.0);  // test aid: BCG_DEBUG_FAIL_ITER
  if (!G2.all_finite()) BCG_FAIL(c, BCG_ERR_NUMERIC, "thinQR: Gram matrix is not finite");
  if (!bcg::cholesky_upper(G2, st->rho)) BCG_FAIL(c, BCG_ERR_NUMERIC, "thinQR: Gram matrix is not positive definite");
  st->delta = st->rho * st->delta;                       // :153
  st->residual = max_ratio(st->delta.row_norms(), st->b_norm);  // :155

  const bool tracing = trace && trace->recorded < trace->capacity;
  double* tm = nullptr;
  double* tr = nullptr;
  const size_t mm2 = static_cast<size_t>(m) * m * 2;
  if (tracing) {
    tm = trace->mats ? trace->mats + static_cast<size_t>(trace->recorded) * (3 + 2 * n_shifts) * mm2 : nullptr;
    tr = trace->res ? trace->res + static_cast<size_t>(trace->recorded) * (1 + n_shifts) : nullptr;
    if (tm) {
      std::memset(tm, 0, sizeof(double) * (3 + 2 * n_shifts) * mm2);
      st->alpha.store(tm);
      st->rho.store(tm + mm2);
      st->delta.store(tm + 2 * mm2);
    }
    if (tr) {
      tr[0] = st->residual;
      for (int s = 0; s < n_shifts; ++s) tr[1 + s] = -1.0;
    }
  }
  // Coefficients of every active shift (host, m x m), then ONE pass over the fields (phase C):
  //   Q <- Q rho^{-1} (:152) ; X_0 += P_0 alpha delta_old (:145) ; P_0 <- P_0 rho^dagger + Q (:158)
  //   X_s += P_s alpha_s (:175) ; P_s <- P_s beta_s rho^dagger + Q (:177)
  const CMat rho_dag = st->rho.adjoint();
  std::vector<CMat> Acoef(1, alpha_delta), Bcoef(1, rho_dag);
  std::vector<bcg_field*> Xa(1, st->X[0]), Pa(1, st->P[0]);
  std::vector<CMat> A_by_shift(n_shifts), B_by_shift(n_shifts);
  const int n_active = st->n_unconverged;
  for (int s = n_active - 1; s > 0; --s) {  // :161
    const CMat beta_s_inv = Identity + (sigma[s] - sigma[0]) * st->alpha +
                            st->alpha * st->rho_old * st->alpha_inv_old * (Identity - st->beta_s[s]) *
                                st->rho_old.adjoint();                                                   // :163-165
    st->beta_s[s] = bcg::inverse_full_pivot(beta_s_inv);                                                 // :166
    st->alpha_s[s] = st->beta_s[s] * st->alpha * st->rho_old * st->alpha_inv_old * st->alpha_s[s];       // :167-168
    const double residual_shift = max_ratio((st->rho * st->alpha_inv * st->alpha_s[s]).row_norms(), st->b_norm);  // :169-172
    Acoef.push_back(st->alpha_s[s]);                                                                     // :175
    Bcoef.push_back(st->beta_s[s] * rho_dag);                                                            // :177
    A_by_shift[s] = Acoef.back();
    B_by_shift[s] = Bcoef.back();
    Xa.push_back(st->X[s]);
    Pa.push_back(st->P[s]);
    if (tm) {
      st->alpha_s[s].store(tm + (3 + s) * mm2);
      st->beta_s[s].store(tm + (3 + n_shifts + s) * mm2);
    }
    if (tr) tr[1 + s] = residual_shift;
    if (residual_shift < st->eps_shifts) --st->n_unconverged;  // :179-181
  }
  const bool lazy = lazy_q_width(c, m);
  const bool next_certain = more_follow && st->residual > st->eps;
  const bool group_open = static_cast<int>(st->pending.size()) + 1 < st->depth && next_certain;
  bool x0_waits = group_open && st->defer_x0 && lazy &&
                  (st->pending.empty() ? st->P0_spare != nullptr : st->pending[0].x0_deferred);
  // the spare-less form (a group of two that has no spare field: capacity mode, or the spare could not be allocated)
  bool x0_backward = false;
  if (group_open && !x0_waits && st->x0_backward && lazy && st->depth == 2 && st->pending.empty() && n_active >= 2) {
    const CMat rinv = bcg::upper_triangular_inverse(st->rho);
    double f2 = 0.0, g2 = 0.0;
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) {
        f2 += std::norm(st->rho(i, j));
        g2 += std::norm(rinv(i, j));
      }
    x0_backward = x0_waits = std::sqrt(f2 * g2) <= c->x0_cond_limit * m;  // (NaN compares false: the plain update)
  }
  // (with shift 0 alone left -- a single system, or the tail of a solve whose shifted systems have converged -- a group
  // is worth opening only for X_0's sake, and only if it can run to three iterations: pair_shifts_depth)
  if (group_open && (n_active >= 2 || (x0_waits && st->depth >= 3))) {
    // shift 0 now, the others in a later iteration's pass (phase_C_multi)
    DeferredIteration d;
    if (x0_waits) {
      // P_0 alone (k_phaseC_p0): the first iteration of a group leaves its P_0 where it is and writes the new one into
      // the spare field; the later ones update in place.  X_0's update joins the closing pass (DeferredX0).
      st->q_rinv = bcg::upper_triangular_inverse(st->rho);
      const CMat* two[2] = {&st->q_rinv, &rho_dag};
      const double2* Md;
      BCG_TRY(upload_mats(c, m, two, 2, &Md));
      bcg_field* const out = (st->pending.empty() && !x0_backward) ? st->P0_spare : st->P[0];
      {
        ProfScope ps(c, "phaseC_p0", row_bytes(st->Q, 3), product_flops(st->Q, 2));
        bcg::launch_phaseC_p0(c->stream, m, rows_of(st->Q), st->Q->d, st->P[0]->d, out->d, Md, c->row_blocks_C);
      }
      BCG_TRY(check_launch(c, "phaseC_p0"));
      if (st->pending.empty() && !x0_backward) {
        st->P0_first = st->P[0];
        st->P[0] = st->P0_spare;
        st->P0_spare = nullptr;
      }
      d.x0_deferred = true;
      d.x0_backward = x0_backward;
      d.A0 = alpha_delta;
      d.R0 = rho_dag;
    } else {
      BCG_TRY(phase_C(c, st->Q, st->rho, Xa.data(), Pa.data(), 1, Acoef, Bcoef, lazy ? &st->q_rinv : nullptr));
    }
    d.Q = st->Q;
    d.rinv = st->q_rinv;
    d.n_active = n_active;
    d.A = A_by_shift;
    d.B = B_by_shift;
    st->pending.push_back(d);
  } else if (!st->pending.empty()) {
    std::vector<DeferredIteration> pend;
    pend.swap(st->pending);  // whatever happens below, these updates are not applied a second time (sbcgrq_flush_pending)
    const int rc = phase_C_multi(c, pend, st->Q, st->rho, st->X.data(), st->P.data(), alpha_delta, rho_dag, n_active,
                                 A_by_shift, B_by_shift, lazy ? &st->q_rinv : nullptr, lazy, nullptr, st->P0_first);
    for (const DeferredIteration& d : pend) {
      if (!st->T) st->T = d.Q;
      else st->Qfree.push_back(d.Q);
    }
    if (st->P0_first) {  // the group's first P_0 has been read for the last time: its field is the next group's spare
      st->P0_spare = st->P0_first;
      st->P0_first = nullptr;
    }
    BCG_TRY(rc);
  } else {
    BCG_TRY(phase_C(c, st->Q, st->rho, Xa.data(), Pa.data(), static_cast<int>(Xa.size()), Acoef, Bcoef,
                    lazy ? &st->q_rinv : nullptr));
  }
  st->q_lazy = lazy;  // from now on the stored Q is un-normalised: Q_true = Q q_rinv
  if (tracing) trace->recorded += 1;
  return BCG_OK;
}

// An iteration failed (thinQR breakdown, a non-finite Gram matrix, a HIP or communication error) while earlier
// iterations' updates of the shifts >= 1 were still waiting for the pass that closes their group: apply them now, so that
// the X_s the caller keeps are those of the last completed iteration for every shift, as in the reference, where every
// shift is current at any point an error could surface (inc/block_solvers.hpp:161-181 run in every iteration).
int sbcgrq_flush_pending(bcg_sbcgrq_state* st) {
  if (st->pending.empty()) return BCG_OK;
  bcg_context* c = st->c;
  const int m = st->m;
  std::vector<DeferredIteration> pend;
  pend.swap(st->pending);
  const DeferredIteration last = pend.back();
  pend.pop_back();
  const bool lazy = lazy_q_width(c, m);
  int rc = BCG_OK;
  if (pend.empty()) {  // one iteration: the ordinary phase C kernel over the shifts >= 1, its rho^-1 applied in registers
    for (int s0 = 1; s0 < last.n_active && rc == BCG_OK;) {
      const int ns = std::min(bcg::phaseC_max_shifts(m, false), last.n_active - s0);
      const CMat unused = CMat::identity(m);  // slot 0 is skipped by the kernel when the block is stored normalised
      std::vector<const CMat*> mats(1, lazy ? &last.rinv : &unused);
      double2* Xp[8];
      double2* Pp[8];
      for (int k = 0; k < ns; ++k) {
        mats.push_back(&last.A[s0 + k]);
        mats.push_back(&last.B[s0 + k]);
        Xp[k] = st->X[s0 + k]->d;
        Pp[k] = st->P[s0 + k]->d;
      }
      const double2* Md;
      rc = upload_mats(c, m, mats.data(), static_cast<int>(mats.size()), &Md);
      if (rc != BCG_OK) break;
      bcg::launch_phaseC(c->stream, m, rows_of(last.Q), last.Q->d, Xp, Pp, ns, Md, lazy ? 2 : 0, c->row_blocks_C);
      rc = check_launch(c, "phaseC");
      s0 += ns;
    }
  } else {
    const CMat none;
    rc = phase_C_multi(c, pend, last.Q, none, st->X.data(), st->P.data(), none, none, last.n_active, last.A, last.B, nullptr,
                       lazy, &last.rinv);
  }
  pend.push_back(last);
  // ... and X_0's, where they waited too: X_0 += P_0^(0) C + sum_k Q_k (rinv_k D_k), over ALL pending iterations (no
  // closing iteration follows: the sum over q_k runs to the last but one, the last one's q only entered the current P_0)
  if (rc == BCG_OK && pend[0].x0_deferred && pend[0].x0_backward) {
    // the spare-less form: one iteration pending, its P_0 gone -- X_0 += (P_0^(1) - q_0) M with the current P_0 (the failed
    // iteration has not touched it) and M = R_0^-1 A_0
    const CMat M = pend[0].rinv.adjoint() * pend[0].A0;
    rc = rmul(c, st->X[0], st->P[0], M, 0.0, bcg::RMUL_ADD, "block_axpy");
    if (rc == BCG_OK) rc = rmul(c, st->X[0], pend[0].Q, -(pend[0].rinv * M), 0.0, bcg::RMUL_ADD, "block_axpy");
  } else if (rc == BCG_OK && st->P0_first && pend[0].x0_deferred) {
    const DeferredX0 x0 = compose_x0(pend);
    rc = rmul(c, st->X[0], st->P0_first, x0.mats[0], 0.0, bcg::RMUL_ADD, "block_axpy");
    for (size_t k = 0; k + 1 < x0.mats.size() && rc == BCG_OK; ++k)
      rc = rmul(c, st->X[0], pend[k].Q, pend[k].rinv * x0.mats[k + 1], 0.0, bcg::RMUL_ADD, "block_axpy");
  }
  if (st->P0_first) {
    st->P0_spare = st->P0_first;
    st->P0_first = nullptr;
  }
  for (const DeferredIteration& d : pend) {
    if (d.Q == st->Q || d.Q == st->T) continue;
    if (!st->T) st->T = d.Q;
    else st->Qfree.push_back(d.Q);
  }
  if (rc == BCG_OK) rc = stream_sync(c);
  return rc;
}

}  // namespace

extern "C" {

int bcg_sbcgrq_begin(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* const* X, bcg_field* B, int n_shifts,
                     const double* sigma, double eps, double eps_shifts, int consume_B, bcg_sbcgrq_state** out) {
  DeviceScope on_device(c);
  if (!c || !g || !X || !B || !sigma || !out || n_shifts < 1 || g->ctx != c || B->ctx != c) return BCG_ERR_INVALID;
  const int m = B->m;
  for (int s = 0; s < n_shifts; ++s)
    if (!same_shape(X[s], B) || X[s] == B) BCG_FAIL(c, BCG_ERR_INVALID, "SBCGrQ: X[s] must be distinct fields of B's width");
  // :97-101
  if (sigma[0] < 0.0) BCG_FAIL(c, BCG_ERR_INVALID, "SBCGrQ: shifts must be zero or positive");
  if (!std::is_sorted(sigma, sigma + n_shifts)) BCG_FAIL(c, BCG_ERR_INVALID, "SBCGrQ: shifts must be in ascending order");
  BCG_TRY(ensure_scratch(c));
  bcg_sbcgrq_state* st = new bcg_sbcgrq_state();
  st->c = c;
  st->g = g;
  st->mass = mass;
  st->m = m;
  st->n_shifts = n_shifts;
  st->sigma.assign(sigma, sigma + n_shifts);
  st->eps = eps;
  st->eps_shifts = eps_shifts;
  st->X.assign(X, X + n_shifts);
  st->B = B;
  st->n_unconverged = n_shifts;                  // :104
  const CMat Identity = CMat::identity(m);       // :106
  st->alpha = st->rho = st->delta = CMat(m);     // :107
  st->alpha_inv = Identity;                      // :108
  st->alpha_inv_old = st->rho_old = CMat(m);
  st->P.assign(n_shifts, nullptr);
#define BEGIN_TRY(call)          \
  do {                           \
    int rc_ = (call);            \
    if (rc_ != BCG_OK) {         \
      sbcgrq_release(st);        \
      delete st;                 \
      return rc_;                \
    }                            \
  } while (0)
  // Every allocation the solve cannot do without comes FIRST and in one stretch -- T, Q (:109), the P_s (:117), then what
  // the operator needs (tmp or the ring, face buffers, partials; the first iteration would otherwise allocate it lazily,
  // and it must come before the optional residual buffers below: a solve that fits without them must not fail because
  // they took the room) -- so that on a lattice divided over ranks the ranks can AGREE on the outcome before the first
  // collective of the solve (the Gram all-reduce inside thinQR, :115): a rank that ran out of memory would otherwise leave
  // its peers waiting in that all-reduce for ever.  With the agreement every rank returns from a failed begin, with
  // nothing allocated, and the caller can try again with a smaller plan (bench.py steps its ladder down this way).
  int alloc_rc = create_like(c, B, &st->T);  // T is overwritten before it is read, so it is not initialised from B
  if (alloc_rc == BCG_OK) {
    if (consume_B) st->Q = B;
    else alloc_rc = create_like(c, B, &st->Q);
  }
  for (int s = 0; s < n_shifts && alloc_rc == BCG_OK; ++s) alloc_rc = create_like(c, B, &st->P[s]);
  if (alloc_rc == BCG_OK) alloc_rc = reserve_operator_scratch(c, B);
  if (c->distributed && c->have_comm && c->comm.allreduce_sum) {
    const std::string why = c->err;
    (void)hipGetLastError();
    double failed_ranks = alloc_rc == BCG_OK ? 0.0 : 1.0;
    int rc_ = BCG_OK;
    *reinterpret_cast<double*>(c->pin_gram) = failed_ranks;
    if (hipMemcpyAsync(c->dev_gram, c->pin_gram, sizeof(double), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc_ = BCG_ERR_HIP;
    if (rc_ == BCG_OK && c->comm.allreduce_sum(c->comm.user, c->dev_gram, 1) != 0) rc_ = BCG_ERR_COMM;
    if (rc_ == BCG_OK && (hipMemcpyAsync(c->pin_gram, c->dev_gram, sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                          hipStreamSynchronize(c->stream) != hipSuccess))
      rc_ = BCG_ERR_HIP;
    if (rc_ == BCG_OK) failed_ranks = *reinterpret_cast<const double*>(c->pin_gram);
    if (alloc_rc == BCG_OK && rc_ != BCG_OK) {
      alloc_rc = rc_;
      c->err = "SBCGrQ: the ranks could not agree on the outcome of their allocations (all-reduce failed)";
    } else if (alloc_rc == BCG_OK && failed_ranks > 0.0) {
      alloc_rc = BCG_ERR_HIP;
      c->err = "SBCGrQ: another rank of the process grid could not allocate the solve's fields (hipErrorOutOfMemory there)";
    } else {
      c->err = why;
    }
  }
  if (alloc_rc != BCG_OK) {
    sbcgrq_release(st);
    delete st;
    return alloc_rc;
  }
  if (!consume_B) BEGIN_TRY(bcg_field_copy(st->Q, B));
  for (int s = 0; s < n_shifts; ++s) BEGIN_TRY(bcg_field_set_zero(X[s]));  // :111-113
  BEGIN_TRY(thin_qr(c, st->Q, st->delta));                                  // :115
  st->rho = st->delta;                                                      // :116
  for (int s = 0; s < n_shifts; ++s) BEGIN_TRY(bcg_field_copy(st->P[s], st->Q));  // :117
#undef BEGIN_TRY
  st->alpha_s.assign(n_shifts, Identity);  // :122
  st->beta_s.assign(n_shifts, Identity);   // :123
  st->depth = pair_shifts_depth(c, m, n_shifts);
  for (int k = 2; k < st->depth; ++k) {  // depth 2 needs none (T doubles as the second residual buffer)
    bcg_field* q = nullptr;
    if (create_like(c, B, &q) != BCG_OK) {  // no room for another residual buffer: a smaller depth
      (void)hipGetLastError();
      c->err.clear();
      st->depth = k;
      break;
    }
    st->Qfree.push_back(q);
  }
  // deferred X_0 (DeferredX0): one more field, optional like the residual buffers
  if (st->depth >= 2 && x0_may_wait(c, m)) {
    if (create_like(c, B, &st->P0_spare) == BCG_OK) {
      st->defer_x0 = true;
    } else {
      (void)hipGetLastError();
      c->err.clear();
      st->P0_spare = nullptr;
    }
  }
  st->x0_backward = st->depth == 2 && !st->defer_x0 && c->defer_x0 && lazy_q_width(c, m) && (m == 8 || m == 16);
  if (n_shifts < 2 && (st->depth < 3 || !st->defer_x0)) {  // a single system groups for X_0's sake or not at all
    for (bcg_field* q : st->Qfree) bcg_field_destroy(q);
    st->Qfree.clear();
    if (st->P0_spare) bcg_field_destroy(st->P0_spare);
    st->P0_spare = nullptr;
    st->defer_x0 = false;
    st->depth = 1;
  }
  st->iter = 0;                            // :126
  st->b_norm = st->delta.row_norms();      // :130
  st->residual = 1.0;                      // :131
  *out = st;
  return BCG_OK;
}

int bcg_sbcgrq_iterate(bcg_sbcgrq_state* st, int max_new_iterations, int* iterations_total, double* residual_out,
                       bcg_sbcgrq_trace* trace) {
  DeviceScope on_device(st ? st->c : nullptr);
  if (!st) return BCG_ERR_INVALID;
  if (st->failed) BCG_FAIL(st->c, BCG_ERR_INVALID, "SBCGrQ: an earlier iteration on this state returned an error");
  int done = 0;
  while (st->residual > st->eps && done < max_new_iterations) {  // :132
    const int rc = sbcgrq_iteration(st, trace, done + 1 < max_new_iterations);
    if (rc != BCG_OK) {
      st->failed = true;
      const std::string why = st->c->err;
      (void)hipGetLastError();
      if (sbcgrq_flush_pending(st) != BCG_OK) st->c->err = why + " (and the deferred updates of the shifted systems could not be applied)";
      else st->c->err = why;
      return rc;
    }
    ++done;
  }
  BCG_TRY(stream_sync(st->c));
  if (iterations_total) *iterations_total = st->iter;
  if (residual_out) *residual_out = st->residual;
  return BCG_OK;
}

int bcg_sbcgrq_end(bcg_sbcgrq_state* st) {
  DeviceScope on_device(st ? st->c : nullptr);
  if (!st) return BCG_OK;
  (void)stream_sync(st->c);
  sbcgrq_release(st);
  delete st;
  return BCG_OK;
}

int bcg_sbcgrq_solve(bcg_context* c, const bcg_gauge* g, double mass, bcg_field* const* X, bcg_field* B, int n_shifts,
                     const double* sigma, double eps, double eps_shifts, int max_iterations, int consume_B,
                     int* iterations_out, double* residual_out, bcg_sbcgrq_trace* trace) {
  DeviceScope on_device(c);
  bcg_sbcgrq_state* st = nullptr;
  if (trace) trace->recorded = 0;
  BCG_TRY(bcg_sbcgrq_begin(c, g, mass, X, B, n_shifts, sigma, eps, eps_shifts, consume_B, &st));
  const int rc = bcg_sbcgrq_iterate(st, max_iterations, iterations_out, residual_out, trace);  // :184
  bcg_sbcgrq_end(st);
  return rc;
}

}  // extern "C"

