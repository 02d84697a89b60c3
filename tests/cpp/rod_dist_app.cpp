// rod_dist_app.cpp -- the domain-decomposed spherocylinder step (BASELINE configs[3]) driven from a C++ host program
// through mundy_hip/stepper.hpp (mech::DistributedSpherocylinderStepper): one process per rank, RCCL transport, no
// Python, no torch, no MPI.  The launcher's only job -- handing the 128-byte RCCL id from rank 0 to the others -- is
// done through a file in <rendezvous_dir> (an MPI host would MPI_Bcast it).
// Usage: rod_dist_app <input.bin> <steps> <rank> <world> <rendezvous_dir> [reuse | migrate <box_edge>]
//   reuse: apply the rebuild rule across ranks instead of rebuilding the neighbour list every step
//   migrate: ownership follows the bodies over a 16^3 Hilbert lattice on [0, box_edge]^3, the curve re-cut by work
//            every third rebalance (DistributedSpherocylinderStepper::rebalance)
//   input.bin: uint64 n, then doubles center[3n] quat[4n] radius[n] length[n] mob_trans[n] mob_rot[n], bodies already in
//   curve order; rank r owns the r-th of `world` equal contiguous ranges and uses device r % device_count.
// Prints one line per step and a bit-level checksum of the rank's final centres / orientations.
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "mundy_hip/stepper.hpp"

using namespace mundy_hip;

static std::vector<double> read_doubles(std::FILE* f, size_t count) {
  std::vector<double> v(count);
  if (std::fread(v.data(), sizeof(double), count, f) != count) {
    std::fprintf(stderr, "short read\n");
    std::exit(2);
  }
  return v;
}
static std::vector<double> rows(const std::vector<double>& a, size_t width, size_t first, size_t count) {
  return std::vector<double>(a.begin() + width * first, a.begin() + width * (first + count));
}
static unsigned long long checksum(const std::vector<double>& v) {  // order-sensitive FNV-1a over the bit patterns
  unsigned long long h = 1469598103934665603ull;
  for (double d : v) {
    unsigned long long b;
    std::memcpy(&b, &d, sizeof b);
    h = (h ^ b) * 1099511628211ull;
  }
  return h;
}

int main(int argc, char** argv) {
  if (argc < 6) {
    std::fprintf(stderr, "Usage: %s <input.bin> <steps> <rank> <world> <rendezvous_dir>\n", argv[0]);
    return 1;
  }
  const int steps = std::atoi(argv[2]), rank = std::atoi(argv[3]), world = std::atoi(argv[4]);
  const std::string dir = argv[5];
  const bool reuse = argc > 6 && std::string(argv[6]) == "reuse";
  const bool migrate = argc > 7 && std::string(argv[6]) == "migrate";
  const double box_edge = migrate ? std::atof(argv[7]) : 0.0;
  std::FILE* f = std::fopen(argv[1], "rb");
  if (!f) {
    std::perror(argv[1]);
    return 2;
  }
  std::uint64_t n = 0;
  if (std::fread(&n, sizeof n, 1, f) != 1) return 2;
  const auto center = read_doubles(f, 3 * n), quat = read_doubles(f, 4 * n), radius = read_doubles(f, n),
             length = read_doubles(f, n), mob_t = read_doubles(f, n), mob_r = read_doubles(f, n);
  std::fclose(f);

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
    std::fprintf(stderr, "no HIP device\n");
    return 3;
  }
  if (hipSetDevice(rank % ndev) != hipSuccess) return 3;

  // rendezvous: rank 0 publishes the id (write + rename, so a reader never sees half a file), the others wait for it
  unsigned char id[MHIP_COMM_ID_BYTES];
  const std::string path = dir + "/rccl_id.bin";
  if (rank == 0) {
    check(mhip_comm_unique_id(id));
    const std::string tmp = path + ".tmp";
    std::FILE* o = std::fopen(tmp.c_str(), "wb");
    if (!o || std::fwrite(id, 1, sizeof id, o) != sizeof id) return 4;
    std::fclose(o);
    if (std::rename(tmp.c_str(), path.c_str()) != 0) return 4;
  } else {
    std::FILE* in = nullptr;
    for (int tries = 0; tries < 6000 && !(in = std::fopen(path.c_str(), "rb")); ++tries) usleep(10000);
    if (!in || std::fread(id, 1, sizeof id, in) != sizeof id) {
      std::fprintf(stderr, "rank %d: no RCCL id at %s\n", rank, path.c_str());
      return 4;
    }
    std::fclose(in);
  }
  mhip_comm_t comm = nullptr;
  check(mhip_comm_create_rccl(&comm, id, rank, world));
  {  // the per-iteration reduction records through the node's mailbox (slots in the ranks' device memory)
    int opened = 0;
    check(mhip_comm_mailbox_open(comm, &opened, nullptr));
    std::printf("MAILBOX rank %d opened %d\n", rank, opened);
    // ... and the per-iteration velocity halo through the inboxes (opened by the first ghost plan; send / recv otherwise)
    check(mhip_comm_halo_ipc_enable(comm, world > 1 ? 1 : 0));
  }

  const size_t base = n / world, rem = n % world;
  const size_t first = rank * base + (static_cast<size_t>(rank) < rem ? rank : rem);
  const size_t mine = base + (static_cast<size_t>(rank) < rem ? 1 : 0);
  convex::PGDConfig<double> cfg;
  cfg.max_iters = 10000;  // NgpLcp.cpp:851-852
  cfg.tol = 1e-5;
  {
    mech::DistributedSpherocylinderStepper st(comm, first, rows(center, 3, first, mine), rows(quat, 4, first, mine),
                                              rows(radius, 1, first, mine), rows(length, 1, first, mine),
                                              rows(mob_t, 1, first, mine), rows(mob_r, 1, first, mine), /*dt=*/5e-3,
                                              /*search_buffer=*/0.1, cfg);
    if (migrate) {
      const double lo[3] = {0.0, 0.0, 0.0}, hi[3] = {box_edge, box_edge, box_edge};
      st.set_domain(lo, hi, /*curve_level=*/4, /*recut_every=*/3);
    }
    for (int k = 0; k < steps; ++k) {
      const auto t0 = std::chrono::steady_clock::now();
      const auto s = st.step(true, /*force_rebuild=*/!reuse, migrate);
      const double ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      std::printf("STEP %d rank %d contacts %zu iterations %u residual %.17g converged %d ghosts %zu interior %zu rebuilt %d ms %.3f\n",
                  k, rank, s.local_contacts, s.num_iters, s.residual, s.converged ? 1 : 0, s.ghosts, s.interior_contacts,
                  s.rebuilt ? 1 : 0, ms);
    }
    // (the vectors are sized to the owned bodies, at least one element)
    auto owned = [&](const DeviceVector& v, size_t w) {
      std::vector<double> h = v.download();
      h.resize(w * st.num_bodies());
      return h;
    };
    std::printf("CHECKSUM rank %d center %016llx quat %016llx entity %016llx owned %zu\n", rank,
                checksum(owned(st.center(), 3)), checksum(owned(st.quat(), 4)), checksum(owned(st.entity_ids(), 1)),
                st.num_bodies());
  }
  check(mhip_comm_destroy(comm));
  return 0;
}
