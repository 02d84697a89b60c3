// rod_step_app.cpp -- the headline workload (BASELINE configs[2]: spherocylinders, frictionless LCP) driven from a C++
// host program through mundy_hip/stepper.hpp, with no Python and no torch in the process.
// Usage: rod_step_app <input.bin> <steps> [reorder_cell] [periodic_box_edge]
//   input.bin: uint64 n, then doubles center[3n] quat[4n] radius[n] length[n] mob_trans[n] mob_rot[n]
// Prints one line per step and a bit-level checksum of the final centres / orientations, so the test can compare the
// whole trajectory with the Python driver's.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
  if (argc < 3) {
    std::fprintf(stderr, "Usage: %s <input.bin> <steps> [reorder_cell] [periodic_box_edge]\n", argv[0]);
    return 1;
  }
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
  const int steps = std::atoi(argv[2]);
  const double cell = argc > 3 ? std::atof(argv[3]) : 0.0;

  convex::PGDConfig<double> cfg;
  cfg.max_iters = 10000;  // NgpLcp.cpp:851-852
  cfg.tol = 1e-5;
  const double edge = argc > 4 ? std::atof(argv[4]) : 0.0;  // > 0: cubic periodic box [0, edge)^3
  const double box[3] = {edge, edge, edge};
  mech::SpherocylinderStepper st(center, quat, radius, length, mob_t, mob_r, /*dt=*/5e-3, /*search_buffer=*/0.1, cfg,
                                 edge > 0.0 ? box : nullptr);
  if (cell > 0.0) {
    const double lo[3] = {0.0, 0.0, 0.0};
    st.reorder_bodies(cell, lo);
  }
  for (int k = 0; k < steps; ++k) {
    const auto t0 = std::chrono::steady_clock::now();
    const mech::StepStats s = st.step(true, false);
    const double ms = 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("STEP %d contacts %zu iterations %u residual %.17g converged %d rebuilt %d ms %.3f\n", k, s.num_contacts,
                s.num_iters, s.residual, s.converged ? 1 : 0, s.rebuilt ? 1 : 0, ms);
  }
  std::printf("CHECKSUM center %016llx quat %016llx\n", checksum(st.center().download()), checksum(st.quat().download()));
  return 0;
}
