// blockcg/fields.hpp -- drop-in for the reference's inc/fields.hpp on MI355X.
//
// Same names, template parameter and member functions as the reference (inc/fields.hpp:18-148):
//   N_f, block_fermion<N_rhs>, fermion, block_matrix<N_rhs>, block_fermion_field<N_rhs>, fermion_field.
// Storage lives in HBM behind the C ABI (include/blockcg_hip.h); every member that loops over lattice
// sites in the reference is one HIP kernel here.  operator[] keeps the reference's host element access
// (benchmark.cpp:61-63) through a lazily synchronised host mirror: reading or writing an element
// downloads the field once, and the next device operation uploads it again if it was written.
//
// The reference's constructor takes only the volume: block_fermion_field<N>(V) is a 1-D lattice of V sites
// (inc/fields.hpp:35, inc/dirac_op.hpp:14-21).  n-D lattices and multi-GPU sub-lattices use
// block_fermion_field<N>(blockcg::lattice&).
#ifndef BLOCKCG_FIELDS_HPP
#define BLOCKCG_FIELDS_HPP
#include <stdlib.h>

#include <complex>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/blockcg_hip.h"
#include "small_matrix.hpp"

namespace blockcg {

inline void check(int rc, const bcg_context* ctx, const char* what) {
  if (rc != BCG_OK) throw std::runtime_error(std::string(what) + ": " + bcg_last_error(ctx));
}

// The reference draws its links and sources from std::rand() (inc/dirac_op.hpp:28-31, inc/fields.hpp:62-66), and
// a driver that calls srand(k) expects the same lattice from these headers.  The GPU runtime, however, consumes
// rand() values of its own while it initialises (observed: B differs from run to run).  While an object of this
// class lives, rand()/random() run on a private state; the caller's sequence continues untouched afterwards
// (glibc's rand() shares random()'s state, which initstate/setstate swap).
class rand_state_guard {
 public:
  rand_state_guard() { prev_ = initstate(0x5eed, scratch(), 256); }
  ~rand_state_guard() { setstate(prev_); }
  rand_state_guard(const rand_state_guard&) = delete;
  rand_state_guard& operator=(const rand_state_guard&) = delete;

 private:
  static char* scratch() {
    static char st[256];
    return st;
  }
  char* prev_;
};

// Allocator of the host mirror behind operator[]: pinned memory (bcg_host_alloc), so that a whole-field download or
// upload runs at the bus rate straight into / out of the mirror; elements are NOT value-initialised on resize (the
// mirror is always filled by a download or by setRandom before it is read).
template <class T>
struct pinned_allocator {
  typedef T value_type;
  pinned_allocator() {}
  template <class U>
  pinned_allocator(const pinned_allocator<U>&) {}
  T* allocate(std::size_t n) {
    void* p = nullptr;
    rand_state_guard keep_callers_rand_sequence;  // the runtime may draw from rand() (see rand_state_guard)
    if (bcg_host_alloc(n * sizeof(T), &p) != BCG_OK || !p) throw std::bad_alloc();
    return static_cast<T*>(p);
  }
  void deallocate(T* p, std::size_t) { bcg_host_free(p); }
  template <class U>
  void construct(U*) {}  // default-initialisation: leave the bytes as they are
  template <class U, class... A>
  void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(static_cast<A&&>(a)...); }
  template <class U>
  bool operator==(const pinned_allocator<U>&) const { return true; }
  template <class U>
  bool operator!=(const pinned_allocator<U>&) const { return false; }
};

// One GPU's (sub-)lattice: owns the bcg_context.
class lattice {
 public:
  explicit lattice(const std::vector<int>& dims, int device = 0, const std::vector<int>& grid = {},
                   const std::vector<int>& coords = {}, void* stream = nullptr)
      : dims_(dims) {
    rand_state_guard keep_callers_rand_sequence;
    check(bcg_context_create(&ctx_, device, stream, static_cast<int>(dims.size()), dims.data(),
                             grid.empty() ? nullptr : grid.data(), coords.empty() ? nullptr : coords.data()),
          nullptr, "bcg_context_create");
    V_ = static_cast<int>(bcg_local_volume(ctx_));
  }
  ~lattice() { bcg_context_destroy(ctx_); }
  lattice(const lattice&) = delete;
  lattice& operator=(const lattice&) = delete;
  bcg_context* ctx() const { return ctx_; }
  int V() const { return V_; }
  const std::vector<int>& dims() const { return dims_; }
  // the reference's implicit lattice: 1-D, V sites, one per process and volume
  static lattice& one_dimensional(int V) {
    static std::map<int, std::unique_ptr<lattice>> cache;
    std::unique_ptr<lattice>& p = cache[V];
    if (!p) p.reset(new lattice(std::vector<int>{V}));
    return *p;
  }

 private:
  std::vector<int> dims_;
  bcg_context* ctx_ = nullptr;
  int V_ = 0;
};

}  // namespace blockcg

constexpr int N_f = 3;  // inc/fields.hpp:18
template <int N_rhs>
using block_fermion = blockcg::cmatrix<N_f, N_rhs>;  // :19-20
typedef block_fermion<1> fermion;                     // :21
template <int N_rhs>
using block_matrix = blockcg::cmatrix<N_rhs, N_rhs>;  // :22-23

template <int N_rhs>
class block_fermion_field {
 public:
  int V;  // :33

  explicit block_fermion_field(int V_) : V(V_), lat_(&blockcg::lattice::one_dimensional(V_)) { alloc(); }  // :35
  explicit block_fermion_field(blockcg::lattice& lat) : V(lat.V()), lat_(&lat) { alloc(); }
  // Half-volume field: the lat.V() / 2 sites of one parity (include/blockcg_hip.h, bcg_field_create_half).  dirac_op::op,
  // every member below and every solver take operands of one parity; see blockcg::SBCGrQ_half_volume.
  block_fermion_field(blockcg::lattice& lat, int parity) : V(lat.V() / 2), lat_(&lat), parity_(parity) { alloc(); }
  int parity() const { return parity_; }  // -1: all sites
  // this (half) <- the sites of its parity of `full`
  void restrict_from(const block_fermion_field& full) {
    full.flush();
    blockcg::check(bcg_field_parity_copy(full.f_, f_, 1), lat_->ctx(), "bcg_field_parity_copy");
    host_valid_ = host_dirty_ = false;
  }
  // this (full): its sites of half's parity <- half
  void insert(const block_fermion_field& half) {
    dev1();
    half.flush();
    blockcg::check(bcg_field_parity_copy(f_, half.f_, 0), lat_->ctx(), "bcg_field_parity_copy");
  }
  block_fermion_field(const block_fermion_field& o) : V(o.V), lat_(o.lat_), parity_(o.parity_) {  // deep copy, value semantics
    alloc();
    o.flush();
    blockcg::rand_state_guard keep_callers_rand_sequence;
    blockcg::check(bcg_field_copy(f_, o.f_), lat_->ctx(), "bcg_field_copy");
  }
  block_fermion_field& operator=(const block_fermion_field& o) {
    if (this != &o) {
      if (o.lat_ != lat_) throw std::runtime_error("block_fermion_field: assignment between lattices");
      o.flush();
      blockcg::check(bcg_field_copy(f_, o.f_), lat_->ctx(), "bcg_field_copy");
      host_valid_ = host_dirty_ = false;
    }
    return *this;
  }
  ~block_fermion_field() { bcg_field_destroy(f_); }

  // [i] returns the site tile on the host (:37-38)
  block_fermion<N_rhs>& operator[](int i) {
    pull();
    host_dirty_ = true;
    return host_[i];
  }
  const block_fermion<N_rhs>& operator[](int i) const {
    pull();
    return host_[i];
  }
  block_fermion_field& operator+=(const block_fermion_field& rhs) {  // :40-46
    dev2(rhs);
    blockcg::check(bcg_field_add_assign(f_, rhs.f_), lat_->ctx(), "operator+=");
    return *this;
  }
  block_fermion_field& operator-=(const block_fermion_field& rhs) {  // :47-53
    dev2(rhs);
    blockcg::check(bcg_field_sub_assign(f_, rhs.f_), lat_->ctx(), "operator-=");
    return *this;
  }
  void setZero() {  // :57-61
    blockcg::check(bcg_field_set_zero(f_), lat_->ctx(), "setZero");
    host_valid_ = host_dirty_ = false;
  }
  // :62-66 -- drawn on the host from std::rand() in the reference's order, then uploaded, so that
  // srand(k) gives the reference's right-hand sides
  void setRandom() {
    host_.resize(V);
    for (int ix = 0; ix < V; ++ix) host_[ix].setRandom();
    host_valid_ = host_dirty_ = true;
  }
  // counter-based generator on the device (value depends on the global element index only); for
  // lattices too large to draw from std::rand()
  void setRandomDevice(unsigned long long seed) {
    blockcg::check(bcg_field_fill_random(f_, seed), lat_->ctx(), "setRandomDevice");
    host_valid_ = host_dirty_ = false;
  }
  // this <- this + rhs * rhs_multiplier (:70-77)
  block_fermion_field& add(const block_fermion_field& rhs, double rhs_multiplier) {
    dev2(rhs);
    blockcg::check(bcg_field_add_scalar(f_, rhs.f_, rhs_multiplier), lat_->ctx(), "add");
    return *this;
  }
  block_fermion_field& add(const block_fermion_field& rhs, const block_matrix<N_rhs>& rhs_multiplier) {
    if (&rhs == this) return add(block_fermion_field(rhs), rhs_multiplier);  // the reference accepts rhs == *this (:74)
    dev2(rhs);
    blockcg::check(bcg_field_add_matrix(f_, rhs.f_, reinterpret_cast<const double*>(rhs_multiplier.data())), lat_->ctx(), "add");
    return *this;
  }
  // this <- this * lhs_multiplier + rhs * rhs_multiplier (:79-90)
  block_fermion_field& rescale_add(double lhs_multiplier, const block_fermion_field& rhs, double rhs_multiplier) {
    dev2(rhs);
    blockcg::check(bcg_field_rescale_add_scalar(f_, lhs_multiplier, rhs.f_, rhs_multiplier), lat_->ctx(), "rescale_add");
    return *this;
  }
  block_fermion_field& rescale_add(const block_matrix<N_rhs>& lhs_multiplier, const block_fermion_field& rhs,
                                   double rhs_multiplier) {
    if (&rhs == this) return rescale_add(lhs_multiplier, block_fermion_field(rhs), rhs_multiplier);
    dev2(rhs);
    blockcg::check(bcg_field_rescale_add_matrix(f_, reinterpret_cast<const double*>(lhs_multiplier.data()), rhs.f_, rhs_multiplier),
                   lat_->ctx(), "rescale_add");
    return *this;
  }
  // The reference templates both multipliers (:69-70, :80-82); the two remaining combinations, which no reference caller
  // uses, are composed from the kernels above (two passes).
  block_fermion_field& rescale_add(double lhs_multiplier, const block_fermion_field& rhs,
                                   const block_matrix<N_rhs>& rhs_multiplier) {
    if (&rhs == this) return rescale_add(lhs_multiplier, block_fermion_field(rhs), rhs_multiplier);
    dev2(rhs);
    blockcg::check(bcg_field_rescale_add_scalar(f_, lhs_multiplier, rhs.f_, 0.0), lat_->ctx(), "rescale_add");  // scale only
    blockcg::check(bcg_field_add_matrix(f_, rhs.f_, reinterpret_cast<const double*>(rhs_multiplier.data())), lat_->ctx(),
                   "rescale_add");
    return *this;
  }
  block_fermion_field& rescale_add(const block_matrix<N_rhs>& lhs_multiplier, const block_fermion_field& rhs,
                                   const block_matrix<N_rhs>& rhs_multiplier) {
    if (&rhs == this) return rescale_add(lhs_multiplier, block_fermion_field(rhs), rhs_multiplier);
    dev2(rhs);
    blockcg::check(bcg_field_rescale_add_matrix(f_, reinterpret_cast<const double*>(lhs_multiplier.data()), rhs.f_, 0.0),
                   lat_->ctx(), "rescale_add");
    blockcg::check(bcg_field_add_matrix(f_, rhs.f_, reinterpret_cast<const double*>(rhs_multiplier.data())), lat_->ctx(),
                   "rescale_add");
    return *this;
  }
  double real_dot(const block_fermion_field<1>& rhs) const {  // :93-99
    flush();
    rhs.flush();
    double r = 0.0;
    blockcg::check(bcg_field_real_dot(f_, rhs.handle(), &r), lat_->ctx(), "real_dot");
    return r;
  }
  block_matrix<N_rhs> hermitian_dot(const block_fermion_field& rhs) const {  // :103-122
    flush();
    rhs.flush();
    block_matrix<N_rhs> R;
    blockcg::check(bcg_field_hermitian_dot(f_, rhs.f_, reinterpret_cast<double*>(R.data())), lat_->ctx(), "hermitian_dot");
    return R;
  }
  block_fermion_field& multiply_upper_triangular_inverse_RHS(const block_matrix<N_rhs>& R) {  // :125-136
    dev1();
    blockcg::check(bcg_field_tri_solve_rhs(f_, reinterpret_cast<const double*>(R.data())), lat_->ctx(), "tri_solve");
    return *this;
  }
  block_fermion_field& thinQR(block_matrix<N_rhs>& R) {  // :140-146
    dev1();
    blockcg::check(bcg_field_thin_qr(f_, reinterpret_cast<double*>(R.data())), lat_->ctx(), "thinQR");
    return *this;
  }

  // --- plumbing used by dirac_op.hpp / block_solvers.hpp
  bcg_field* handle() const { return f_; }
  blockcg::lattice& lat() const { return *lat_; }
  void flush() const {  // make the device copy current
    if (host_dirty_) {
      blockcg::rand_state_guard keep_callers_rand_sequence;
      blockcg::check(bcg_field_upload(f_, reinterpret_cast<const double*>(host_.data())), lat_->ctx(), "upload");
      host_dirty_ = false;
    }
  }
  void device_written() { host_valid_ = host_dirty_ = false; }  // a kernel overwrote the device copy

 private:
  void alloc() {
    blockcg::rand_state_guard keep_callers_rand_sequence;
    if (parity_ >= 0) blockcg::check(bcg_field_create_half(lat_->ctx(), N_rhs, parity_, &f_), lat_->ctx(), "bcg_field_create_half");
    else blockcg::check(bcg_field_create(lat_->ctx(), N_rhs, &f_), lat_->ctx(), "bcg_field_create");
  }
  void pull() const {
    if (!host_valid_) {
      host_.resize(V);
      blockcg::check(bcg_field_download(f_, reinterpret_cast<double*>(host_.data())), lat_->ctx(), "download");
      host_valid_ = true;
    }
  }
  void dev1() {
    flush();
    host_valid_ = false;
  }
  void dev2(const block_fermion_field& rhs) {
    dev1();
    rhs.flush();
  }
  blockcg::lattice* lat_;
  int parity_ = -1;
  bcg_field* f_ = nullptr;
  mutable std::vector<block_fermion<N_rhs>, blockcg::pinned_allocator<block_fermion<N_rhs>>> host_;
  mutable bool host_valid_ = false, host_dirty_ = false;
};
typedef block_fermion_field<1> fermion_field;  // :148

#endif
