// blockcg/dirac_op.hpp -- drop-in for the reference's inc/dirac_op.hpp on MI355X.
//
// Same class name and public shape (inc/dirac_op.hpp:8-44): explicit dirac_op(int V, double mass = 0.1),
// public V and mass, template <int N_rhs> void op(lhs, rhs) const.  The 1-D constructor draws its links
// from std::rand() exactly as the reference does (:27-32), so srand(k) reproduces the reference's operator.
// The n-D constructor takes a blockcg::lattice; its links come from the device generator or from the caller.
#ifndef BLOCKCG_DIRAC_OP_HPP
#define BLOCKCG_DIRAC_OP_HPP
#include "fields.hpp"

class dirac_op {
 public:
  int V;        // :25
  double mass;  // :26

  explicit dirac_op(int V_, double mass_ = 0.1) : V(V_), mass(mass_), lat_(nullptr) {
    std::vector<blockcg::cmatrix<N_f, N_f>> U(V);
    for (int ix = 0; ix < V; ++ix) U[ix].setRandom();  // :28-31, drawn before anything touches the GPU runtime
    lat_ = &blockcg::lattice::one_dimensional(V_);
    create();
    blockcg::rand_state_guard keep_callers_rand_sequence;
    blockcg::check(bcg_gauge_upload(g_.get(), reinterpret_cast<const double*>(U.data())), lat_->ctx(), "bcg_gauge_upload");
  }
  // n-D: links i.i.d. uniform [-1,1) from the counter-based device generator
  dirac_op(blockcg::lattice& lat, double mass_, unsigned long long seed) : V(lat.V()), mass(mass_), lat_(&lat) {
    create();
    blockcg::check(bcg_gauge_fill_random(g_.get(), seed), lat_->ctx(), "bcg_gauge_fill_random");
  }
  // n-D: links given by the caller, [site][mu][3x3 column-major]
  dirac_op(blockcg::lattice& lat, double mass_, const std::complex<double>* links) : V(lat.V()), mass(mass_), lat_(&lat) {
    create();
    blockcg::check(bcg_gauge_upload(g_.get(), reinterpret_cast<const double*>(links)), lat_->ctx(), "bcg_gauge_upload");
  }
  // The reference's dirac_op is implicitly copyable (its links are a std::vector, inc/dirac_op.hpp:10-11).  The links are
  // immutable after construction, so copies here share the device links (the last copy frees them).
  ~dirac_op() = default;
  dirac_op(const dirac_op&) = default;
  dirac_op& operator=(const dirac_op&) = default;

  // lhs = (m^2 - D^2) rhs  (:36-43)
  template <int N_rhs>
  void op(block_fermion_field<N_rhs>& lhs, const block_fermion_field<N_rhs>& rhs) const {
    rhs.flush();
    blockcg::check(bcg_dirac_apply(lat_->ctx(), g_.get(), mass, lhs.handle(), rhs.handle()), lat_->ctx(), "dirac_op::op");
    lhs.device_written();
  }

  bcg_gauge* handle() const { return g_.get(); }
  blockcg::lattice& lat() const { return *lat_; }

 private:
  void create() {
    blockcg::rand_state_guard keep_callers_rand_sequence;
    bcg_gauge* g = nullptr;
    blockcg::check(bcg_gauge_create(lat_->ctx(), &g), lat_->ctx(), "bcg_gauge_create");
    g_ = std::shared_ptr<bcg_gauge>(g, [](bcg_gauge* p) { bcg_gauge_destroy(p); });
  }
  blockcg::lattice* lat_;
  std::shared_ptr<bcg_gauge> g_;
};

#endif
