// stepper.hpp -- the per-timestep contact hot path for spherocylinders as a C++ host loop over the C ABI: what a
// MundyMech-style timestep loop would hold (the reference's only such loops are its scrap apps,
// scrap/lcp_spheres/NgpLcp.cpp:835-920; scrap/.../Bacteria.cpp:1013-1110).  Host code only sequences kernels:
//   [Z-order reorder] -> compute_aabb -> GenNeighborLinks::generate -> segment records -> contact_spherocylinders
//   -> ContactOperator (rod-compressed) -> solve_lcp (fused BBPGD) -> body velocities -> Euler + quaternion update
// All arrays stay on the device; the only host reads are the pair count and the solver's convergence polls.
#pragma once
#include <chrono>
#include <memory>

#include "mundy_hip/adapter.hpp"

namespace mundy_hip {
namespace mech {

/// grow-only workspace: reallocates only when the buffer is too small, so a time loop stops allocating after its
/// first steps (hipMalloc / hipFree cost more than most kernels of a step)
template <class T>
inline T* workspace(DeviceArray<T>& a, size_t n) {
  if (a.size() < n) a = DeviceArray<T>(n + n / 8 + 16);
  return a.data();
}

struct StepStats {
  size_t num_contacts = 0;
  unsigned num_iters = 0;
  double residual = 0.0;
  bool converged = false, rebuilt = false;
};

class SpherocylinderStepper {
 public:
  /// host arrays: center [n][3], quat [n][4] (w, x, y, z), radius [n], length [n], mob_trans [n], mob_rot [n]
  SpherocylinderStepper(const std::vector<double>& center, const std::vector<double>& quat,
                        const std::vector<double>& radius, const std::vector<double>& length,
                        const std::vector<double>& mob_trans, const std::vector<double>& mob_rot, double dt,
                        double search_buffer, convex::PGDConfig<double> cfg, const double* periodic_box = nullptr)
      : n_(radius.size()), dt_(dt), cfg_(cfg), center_(center), quat_(quat), radius_(radius), length_(length),
        mob_t_(mob_trans), mob_r_(mob_rot), brad_(n_), aabb_(6 * n_), seg_(8 * n_), tmp_(4 * n_), perm_(n_) {
    check(mhip_bounding_radius_spherocylinders(n_, radius_.data(), length_.data(), brad_.data(), nullptr));
    links_.set_search_buffer(search_buffer).set_search_kind(MHIP_SEARCH_AABB);
    if (periodic_box) {  // orthorhombic periodic box [0, L): periodic search, nearest-image contacts, wrap_rigid
      periodic_ = true;
      for (int k = 0; k < 3; ++k) box_[k] = periodic_box[k];
      links_.set_periodic_box(box_[0], box_[1], box_[2]);
    }
    links_.concretize();
  }

  /// Z-order permutation of every per-body array by centre
  void reorder_bodies(double cell_size, const double lo[3]) {
    check(mhip_morton_order(n_, center_.data(), lo, cell_size, perm_.data(), nullptr));
    gather(center_, 3);
    gather(quat_, 4);
    gather(radius_, 1);
    gather(length_, 1);
    gather(brad_, 1);
    gather(mob_t_, 1);
    gather(mob_r_, 1);
    // the neighbour list and the operator's incidence index are in the old numbering
    links_.invalidate();
    op_.reset();
  }

  StepStats step(bool integrate = true, bool force_rebuild = false) {
    StepStats st;
    check(mhip_compute_aabb_spherocylinders(n_, center_.data(), quat_.data(), radius_.data(), length_.data(),
                                            aabb_.data(), nullptr));
    st.rebuilt = links_.generate(n_, aabb_.data(), center_.data(), brad_.data(), nullptr, force_rebuild);
    if (st.rebuilt) links_.links_into(pairs_);
    const size_t C = links_.num_links();
    st.num_contacts = C;
    check(mhip_spherocylinder_segments(n_, center_.data(), quat_.data(), radius_.data(), length_.data(), seg_.data(),
                                       nullptr));
    double *sep = workspace(w_sep_, C), *normal = workspace(w_normal_, 3 * C), *s = workspace(w_s_, C),
           *t = workspace(w_t_, C);
    if (periodic_)
      check(mhip_contact_spherocylinders_periodic(C, pairs_.data(), seg_.data(), center_.data(), box_, sep, normal,
                                                  nullptr, nullptr, nullptr, nullptr, s, t, nullptr));
    else
      check(mhip_contact_spherocylinders(C, pairs_.data(), seg_.data(), nullptr, sep, normal, nullptr, nullptr,
                                         nullptr, nullptr, s, t, nullptr));
    // the operator follows the contact list: rebuilt with it, otherwise only its geometry is refreshed
    if (st.rebuilt || !op_)
      op_.reset(new ContactOperator(C, n_, pairs_.data(), normal, ContactOperator::Rods{s, t, seg_.data()},
                                    mob_t_.data(), mob_r_.data(), dt_, nullptr, /*priority=*/sep));
    else
      op_->refresh(normal, ContactOperator::Rods{s, t, seg_.data()});
    ContactOperator& op = *op_;
    double *x = workspace(lambda_, C), *g = workspace(w_g_, C), *x_tmp = workspace(w_xt_, C),
           *g_tmp = workspace(w_gt_, C);
    check(mhip_fill(C, x, 0.0, nullptr));  // lambda = 0 (NgpLcp.cpp:890-891)
    num_lambda_ = C;
    const mhip_space lcp{MHIP_SPACE_LOWER_BOUND, 0.0, 0.0};
    const mhip_pgd_config pc{cfg_.max_iters, cfg_.tol, MHIP_RESIDUAL_PROJECTED_DIFF};
    mhip_solve_result res{};
    check(mhip_bbpgd_solve_contact(op.handle(), sep, &lcp, &pc, x, g, x_tmp, g_tmp, &res, nullptr));
    st.num_iters = res.num_iters;
    st.residual = res.residual;
    st.converged = res.converged != 0;
    if (integrate) {
      const double* vel = nullptr;
      check(mhip_contact_op_body_velocity(op.handle(), &vel));
      check(mhip_integrate_euler(n_, dt_, vel, center_.data(), quat_.data(), nullptr));
      // wrap_rigid_inplace(Spherocylinder): the centre goes back into the box (periodicity.hpp:1094-1113)
      if (periodic_) check(mhip_wrap_rigid(n_, box_, center_.data(), nullptr));
    }
    check(mhip_stream_synchronize(nullptr));
    return st;
  }

  size_t num_bodies() const { return n_; }
  const DeviceVector& center() const { return center_; }
  const DeviceVector& quat() const { return quat_; }
  /// multipliers of the last step: the first num_lambda() entries (the buffer only ever grows)
  const DeviceVector& lambda() const { return lambda_; }
  size_t num_lambda() const { return num_lambda_; }
  const DeviceArray<int32_t>& pairs() const { return pairs_; }

 private:
  void gather(DeviceVector& a, size_t width) {
    check(mhip_gather_rows(n_, width, perm_.data(), a.data(), tmp_.data(), nullptr));
    check(mhip_deep_copy(width * n_, a.data(), tmp_.data(), nullptr));
  }
  size_t n_;
  double dt_;
  bool periodic_ = false;
  double box_[3] = {0.0, 0.0, 0.0};
  convex::PGDConfig<double> cfg_;
  DeviceVector center_, quat_, radius_, length_, mob_t_, mob_r_, brad_, aabb_, seg_, tmp_, lambda_;
  DeviceVector w_sep_, w_normal_, w_s_, w_t_, w_g_, w_xt_, w_gt_;  // per-step workspaces (grow-only)
  size_t num_lambda_ = 0;
  DeviceArray<int32_t> perm_, pairs_;
  mesh::GenNeighborLinks links_;
  std::unique_ptr<ContactOperator> op_;
};

// One rank's share of a spherocylinder system cut along a space-filling curve: this rank owns the bodies
// with global ids [gid_first, gid_first + n), ranks own increasing ranges.  step() =
//   compute_aabb(owned) -> ghost plan + body-record exchange (coarse_search(comm) + change_ghosting,
//   GenNeighborLinkers.hpp:658, :687-711) -> neighbour list over owned + ghosts, ghost-ghost pairs dropped, interior
//   contacts first -> contacts -> rod operator with the owned range -> domain-decomposed BBPGD (velocity halo + 3-double
//   all-gather per iteration, NGPSpheresLCP.cpp:371, :450-452) -> Euler update of the owned bodies.
// The communicator is the caller's (mhip_comm_create_rccl with an id the launcher hands round, or a host transport).
class DistributedSpherocylinderStepper {
 public:
  // gid, centre 3, quaternion 4, radius, length, translational / rotational mobility, entity id (what a body keeps when
  // it changes owner; the gid is its position in the current ownership order)
  static constexpr size_t kRecord = 13;

  DistributedSpherocylinderStepper(mhip_comm_t comm, size_t gid_first, const std::vector<double>& center,
                                   const std::vector<double>& quat, const std::vector<double>& radius,
                                   const std::vector<double>& length, const std::vector<double>& mob_trans,
                                   const std::vector<double>& mob_rot, double dt, double search_buffer,
                                   convex::PGDConfig<double> cfg)
      : comm_(comm), n_(radius.size()), dt_(dt), buffer_(search_buffer), cfg_(cfg), center_(center), quat_(quat),
        radius_(radius), length_(length), mob_t_(mob_trans), mob_r_(mob_rot), aabb_(6 * n_), rec_(kRecord * n_) {
    std::vector<double> gid(n_);
    for (size_t i = 0; i < n_; ++i) gid[i] = static_cast<double>(gid_first + i);
    gid_ = DeviceVector(gid);
    entity_ = DeviceVector(gid);  // until set_entity_ids: a body's id is its first global position
    links_.set_search_buffer(search_buffer).set_search_kind(MHIP_SEARCH_AABB).concretize();
  }

  /// ids the bodies keep for life (integer-valued, below 2^40); default: the global position at construction
  void set_entity_ids(const std::vector<double>& ids) {
    if (ids.size() != n_) throw std::invalid_argument("set_entity_ids: one id per owned body");
    entity_ = DeviceVector(ids);
  }
  /// The lattice the ownership is cut on: (2^level)^3 cells over [lo, hi], visited along mundy::math::hilbert_3d
  /// (Hilbert.hpp:48-83).  Needed by rebalance() / step(..., migrate = true).
  void set_domain(const double lo[3], const double hi[3], int curve_level = 4, int recut_every = 4) {
    for (int a = 0; a < 3; ++a) {
      dom_lo_[a] = lo[a];
      dom_hi_[a] = hi[a];
    }
    level_ = curve_level;
    recut_every_ = recut_every;
    const size_t side = static_cast<size_t>(1) << level_;
    std::vector<int32_t> table(side * side * side);
    check(mhip_hilbert_key_table(level_, table.data()));
    key_table_ = DeviceArray<int32_t>(table);
    have_domain_ = true;
  }

  struct MigrateStats {
    size_t sent = 0, received = 0, owned = 0;
    bool recut = false;
  };
  /// Ownership follows the bodies (replaces stk::balance::balanceStkMesh, NGPSpheresLCP.cpp:956, called every
  /// load_balance_frequency steps, Bacteria.cpp:1076-1078): every owned body moves to the rank that owns its lattice
  /// cell; recut first re-cuts the curve at equal work (1 + contacts per body in the last step).  The owned set ends up
  /// in (cell, entity id) order -- the order a single rank would hold it in.  Collective.
  MigrateStats rebalance(bool recut = true) {
    if (!have_domain_) throw std::runtime_error("rebalance() needs set_domain(lo, hi, level) first");
    int rank = 0, world = 1;
    check(mhip_comm_info(comm_, &rank, &world, nullptr));
    DeviceArray<uint32_t> keys(n_ ? n_ : 1);
    check(mhip_curve_keys(n_, center_.data(), dom_lo_, dom_hi_, level_, key_table_.data(), keys.data(), nullptr));
    MigrateStats ms;
    if (recut || splitters_.size() + 1 != static_cast<size_t>(world)) {
      splitters_.assign(world > 1 ? static_cast<size_t>(world - 1) : 1, 0);
      const size_t ncell = static_cast<size_t>(1) << (3 * level_);
      check(mhip_curve_cut(comm_, n_, keys.data(), weight_.size() == n_ && n_ ? weight_.data() : nullptr, ncell,
                           splitters_.data(), nullptr));
      splitters_.resize(static_cast<size_t>(world - 1));
      ms.recut = true;
    }
    pack_records();
    size_t n_new = 0;
    const int64_t none = 0;
    check(mhip_migrate_plan(comm_, n_, keys.data(), world > 1 ? splitters_.data() : &none, &n_new, &ms.sent, &ms.received,
                            nullptr));
    DeviceVector arrived(kRecord * (n_new ? n_new : 1)), sorted(kRecord * (n_new ? n_new : 1));
    check(mhip_migrate_exchange(comm_, kRecord, rec_.data(), arrived.data(), nullptr));
    // (cell, entity id) order
    DeviceVector c_new(3 * (n_new ? n_new : 1)), e_new(n_new ? n_new : 1);
    check(mhip_copy_strided(n_new, 3, arrived.data() + 1, kRecord, c_new.data(), 3, nullptr));
    check(mhip_copy_strided(n_new, 1, arrived.data() + 12, kRecord, e_new.data(), 1, nullptr));
    DeviceArray<uint32_t> k_new(n_new ? n_new : 1);
    DeviceArray<uint64_t> k64(n_new ? n_new : 1);
    DeviceArray<int32_t> perm(n_new ? n_new : 1);
    check(mhip_curve_keys(n_new, c_new.data(), dom_lo_, dom_hi_, level_, key_table_.data(), k_new.data(), nullptr));
    check(mhip_compose_keys_u64(n_new, k_new.data(), e_new.data(), 40, k64.data(), nullptr));
    check(mhip_sort_by_key_u64(n_new, k64.data(), perm.data(), nullptr));
    check(mhip_gather_rows(n_new, kRecord, perm.data(), arrived.data(), sorted.data(), nullptr));
    // the new owned set, global positions = after the lower ranks' bodies
    n_ = n_new;
    double mine = static_cast<double>(n_);
    DeviceVector sizes(1 + static_cast<size_t>(world));
    check(mhip_memcpy_h2d(sizes.data(), &mine, sizeof mine, nullptr));
    check(mhip_comm_all_gather(comm_, sizes.data(), 1, sizes.data() + 1, nullptr));
    std::vector<double> all(static_cast<size_t>(world));
    check(mhip_memcpy_d2h(all.data(), sizes.data() + 1, all.size() * sizeof(double), nullptr));
    double gid_first = 0.0;
    for (int r = 0; r < rank; ++r) gid_first += all[static_cast<size_t>(r)];
    const size_t cap = n_ ? n_ : 1;
    gid_ = DeviceVector(cap); center_ = DeviceVector(3 * cap); quat_ = DeviceVector(4 * cap); radius_ = DeviceVector(cap);
    length_ = DeviceVector(cap); mob_t_ = DeviceVector(cap); mob_r_ = DeviceVector(cap); entity_ = DeviceVector(cap);
    aabb_ = DeviceVector(6 * cap); rec_ = DeviceVector(kRecord * cap);
    check(mhip_fill_sequence(n_, gid_first, gid_.data(), nullptr));
    struct { DeviceVector* v; size_t w, col; } out[] = {{&center_, 3, 1}, {&quat_, 4, 4}, {&radius_, 1, 8}, {&length_, 1, 9},
                                                        {&mob_t_, 1, 10}, {&mob_r_, 1, 11}, {&entity_, 1, 12}};
    for (const auto& f : out) check(mhip_copy_strided(n_, f.w, sorted.data() + f.col, kRecord, f.v->data(), f.w, nullptr));
    check(mhip_stream_synchronize(nullptr));
    // everything indexed by the old numbering goes
    links_.invalidate();
    op_.reset();
    lay_ = mhip_ghost_layout{};
    weight_ = DeviceVector();
    ++rebalances_;
    ms.owned = n_;
    return ms;
  }

  struct DistStats : StepStats {
    size_t ghosts = 0, local_contacts = 0, interior_contacts = 0, owned_contacts = 0;
    size_t migrated_out = 0, migrated_in = 0;
  };

  /// force_rebuild = false applies the reference's rebuild rule across the ranks: the ghosts' current state travels
  /// through the plan of the last rebuild, every rank tests its local bodies (owned + ghosts) against half the search
  /// buffer (GenNeighborLinkers.hpp:603-615), one all-gather of the flags decides for everybody; without a rebuild the
  /// ghost layout, the partitioned pair list and the operator's incidence index are kept.
  DistStats step(bool integrate = true, bool force_rebuild = true, bool migrate = false) {
    DistStats st;
    if (migrate && (force_rebuild || !op_)) {  // bodies change owner at a rebuild only (the lists are rebuilt anyway)
      const bool recut = splitters_.empty() || (recut_every_ > 0 && rebalances_ % static_cast<unsigned>(recut_every_) == 0);
      const MigrateStats ms = rebalance(recut);
      st.migrated_out = ms.sent;
      st.migrated_in = ms.received;
    }
    // the owned fields interleaved into records (what travels to the ranks that hold these bodies as ghosts)
    const struct { const DeviceVector* v; size_t w; } fields[] = {{&gid_, 1},    {&center_, 3}, {&quat_, 4},  {&radius_, 1},
                                                                  {&length_, 1}, {&mob_t_, 1},  {&mob_r_, 1}, {&entity_, 1}};
    pack_records();
    auto exchange_and_split = [&]() {  // records through the current plan, then split into fields again
      const size_t nl = lay_.num_ghost_lo + n_ + lay_.num_ghost_hi;
      double* local = workspace(w_local_, kRecord * nl);
      check(mhip_ghost_exchange(comm_, kRecord, rec_.data(), local, nullptr));
      double* l_field[8] = {workspace(l_gid_, nl),    workspace(l_center_, 3 * nl), workspace(l_quat_, 4 * nl),
                            workspace(l_radius_, nl), workspace(l_length_, nl),     workspace(l_mt_, nl),
                            workspace(l_mr_, nl),     workspace(l_entity_, nl)};
      size_t c0 = 0;
      for (size_t k = 0; k < 8; ++k) {
        check(mhip_copy_strided(nl, fields[k].w, local + c0, kRecord, l_field[k], fields[k].w, nullptr));
        c0 += fields[k].w;
      }
      return nl;
    };
    bool reuse = false;
    size_t nl = 0;
    if (!force_rebuild && op_) {
      nl = exchange_and_split();
      double flag = links_.objects_moved_too_much(nl, l_center_.data()) ? 1.0 : 0.0;
      int world = 1;
      check(mhip_comm_info(comm_, nullptr, &world, nullptr));
      double* d = workspace(w_flags_, 1 + static_cast<size_t>(world));
      check(mhip_memcpy_h2d(d, &flag, sizeof flag, nullptr));
      check(mhip_comm_all_gather(comm_, d, 1, d + 1, nullptr));
      std::vector<double> all(static_cast<size_t>(world));
      check(mhip_memcpy_d2h(all.data(), d + 1, all.size() * sizeof(double), nullptr));
      reuse = true;
      for (double f : all) reuse = reuse && f == 0.0;
    }
    if (!reuse) {
      check(mhip_compute_aabb_spherocylinders(n_, center_.data(), quat_.data(), radius_.data(), length_.data(),
                                              aabb_.data(), nullptr));
      check(mhip_ghost_plan(comm_, n_, aabb_.data(), buffer_, &lay_, nullptr));
      nl = exchange_and_split();
    }
    const size_t n_lo = lay_.num_ghost_lo;
    st.ghosts = nl - n_;
    st.rebuilt = !reuse;
    double *l_center = l_center_.data(), *l_quat = l_quat_.data(), *l_radius = l_radius_.data(),
           *l_length = l_length_.data();
    double* seg = workspace(w_seg_, 8 * nl);
    if (!reuse) {
      // neighbour list over owned + ghosts; ghost-ghost pairs dropped, interior contacts first
      double *l_aabb = workspace(w_aabb_, 6 * nl), *l_brad = workspace(w_brad_, nl);
      check(mhip_compute_aabb_spherocylinders(nl, l_center, l_quat, l_radius, l_length, l_aabb, nullptr));
      check(mhip_bounding_radius_spherocylinders(nl, l_radius, l_length, l_brad, nullptr));
      links_.generate(nl, l_aabb, l_center, l_brad, nullptr, /*force=*/true);
      const int32_t* all_pairs = links_.links_into(w_all_pairs_);
      const size_t c_all = links_.num_links();
      int32_t* pairs = workspace(w_pairs_, 2 * c_all + 2);
      unsigned char* counted = workspace(w_counted_, c_all + 1);
      check(mhip_partition_pairs_owned(c_all, all_pairs, n_lo, n_, pairs, counted, &n_int_, &num_contacts_, nullptr));
      // work per owned body for the next re-cut of the curve: 1 + its contacts
      if (have_domain_) {
        weight_ = DeviceVector(n_ ? n_ : 1);
        check(mhip_body_work_weights(num_contacts_, pairs, n_lo, n_, weight_.data(), nullptr));
      }
    }
    const size_t C = num_contacts_;
    const int32_t* pairs = w_pairs_.data();
    st.num_contacts = st.local_contacts = C;
    st.interior_contacts = n_int_;
    check(mhip_spherocylinder_segments(nl, l_center, l_quat, l_radius, l_length, seg, nullptr));
    double *sep = workspace(w_sep_, C), *normal = workspace(w_normal_, 3 * C), *s = workspace(w_s_, C),
           *t = workspace(w_t_, C);
    check(mhip_contact_spherocylinders(C, pairs, seg, nullptr, sep, normal, nullptr, nullptr, nullptr, nullptr, s, t,
                                       nullptr));
    if (reuse)
      op_->refresh(normal, ContactOperator::Rods{s, t, seg});
    else
      op_.reset(new ContactOperator(C, nl, pairs, normal, ContactOperator::Rods{s, t, seg}, l_mt_.data(), l_mr_.data(),
                                    dt_, nullptr, /*priority=*/sep));
    ContactOperator& op = *op_;
    double* vel = workspace(w_vel_, 6 * nl);
    check(mhip_fill(6 * nl, vel, 0.0, nullptr));
    check(mhip_contact_op_set_partition(op.handle(), n_lo, n_, w_counted_.data(), vel));
    lay_.halo.velocity = vel;
    double *x = workspace(lambda_, C), *g = workspace(w_g_, C), *x_tmp = workspace(w_xt_, C),
           *g_tmp = workspace(w_gt_, C);
    check(mhip_fill(C, x, 0.0, nullptr));
    num_lambda_ = C;
    const mhip_space lcp{MHIP_SPACE_LOWER_BOUND, 0.0, 0.0};
    const mhip_pgd_config pc{cfg_.max_iters, cfg_.tol, MHIP_RESIDUAL_PROJECTED_DIFF};
    mhip_solve_result res{};
    check(mhip_bbpgd_solve_contact_distributed(op.handle(), comm_, &lay_.halo, n_int_, sep, &lcp, &pc, x, g, x_tmp,
                                               g_tmp, /*poll_every=*/64, &res, nullptr, nullptr));
    st.num_iters = res.num_iters;
    st.residual = res.residual;
    st.converged = res.converged != 0;
    if (integrate) {
      const double* v = nullptr;
      check(mhip_contact_op_body_velocity(op.handle(), &v));
      check(mhip_integrate_euler(n_, dt_, v + 6 * n_lo, l_center + 3 * n_lo, l_quat + 4 * n_lo, nullptr));
      check(mhip_deep_copy(3 * n_, center_.data(), l_center + 3 * n_lo, nullptr));
      check(mhip_deep_copy(4 * n_, quat_.data(), l_quat + 4 * n_lo, nullptr));
    }
    check(mhip_stream_synchronize(nullptr));
    return st;
  }

  size_t num_bodies() const { return n_; }
  const DeviceVector& center() const { return center_; }
  const DeviceVector& quat() const { return quat_; }
  const DeviceVector& entity_ids() const { return entity_; }
  const std::vector<int64_t>& splitters() const { return splitters_; }
  /// multipliers of the last step: the first num_lambda() entries (the buffer only ever grows)
  const DeviceVector& lambda() const { return lambda_; }
  size_t num_lambda() const { return num_lambda_; }

 private:
  void pack_records() {
    const struct { const DeviceVector* v; size_t w; } fields[] = {{&gid_, 1},    {&center_, 3}, {&quat_, 4},  {&radius_, 1},
                                                                  {&length_, 1}, {&mob_t_, 1},  {&mob_r_, 1}, {&entity_, 1}};
    size_t col = 0;
    for (const auto& f : fields) {
      check(mhip_copy_strided(n_, f.w, f.v->data(), f.w, rec_.data() + col, kRecord, nullptr));
      col += f.w;
    }
  }
  mhip_comm_t comm_;
  size_t n_;
  double dt_, buffer_;
  convex::PGDConfig<double> cfg_;
  DeviceVector center_, quat_, radius_, length_, mob_t_, mob_r_, aabb_, rec_, gid_, entity_, weight_, lambda_;
  // ownership lattice (set_domain) and the current cuts of the curve
  bool have_domain_ = false;
  double dom_lo_[3] = {0, 0, 0}, dom_hi_[3] = {1, 1, 1};
  int level_ = 4, recut_every_ = 4;
  unsigned rebalances_ = 0;
  DeviceArray<int32_t> key_table_;
  std::vector<int64_t> splitters_;
  // per-step workspaces (grow-only): local records and fields, geometry, contacts, solver vectors
  DeviceVector w_local_, l_gid_, l_center_, l_quat_, l_radius_, l_length_, l_mt_, l_mr_, l_entity_, w_aabb_, w_brad_, w_seg_,
      w_sep_, w_normal_, w_s_, w_t_, w_vel_, w_g_, w_xt_, w_gt_;
  DeviceVector w_flags_;
  DeviceArray<int32_t> w_pairs_, w_all_pairs_;
  DeviceArray<unsigned char> w_counted_;
  size_t num_lambda_ = 0, n_int_ = 0, num_contacts_ = 0;
  mhip_ghost_layout lay_{};  // of the last rebuild; its lists live in the communicator until the next plan
  mesh::GenNeighborLinks links_;
  std::unique_ptr<ContactOperator> op_;
};

}  // namespace mech
}  // namespace mundy_hip
