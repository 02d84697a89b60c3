// ngp_lcp_app.cpp -- the reference's single-device LCP sphere app (scrap/lcp_spheres/NgpLcp.cpp:835-920) as a C++ host
// loop over the C ABI: random spheres -> neighbour pairs (radius + search buffer) -> signed separation + contact normal
// -> resolve_collisions (the app's own BBPGD variant) -> Euler step, printing what the original prints.
// Usage: ngp_lcp_app <box_size> <num_spheres> [steps]     (same two arguments as the original, plus a step count)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mundy_hip/adapter.hpp"

using namespace mundy_hip;

static double u01(unsigned long long seed, unsigned long long i, unsigned long long s) {  // counter-based, as Philox(seed, i)
  unsigned long long z = seed * 0xD1342543DE82EF95ull + s;
  auto mix = [](unsigned long long v) {
    v += 0x9E3779B97F4A7C15ull;
    v = (v ^ (v >> 30)) * 0xBF58476D1CE4E5B9ull;
    v = (v ^ (v >> 27)) * 0x94D049BB133111EBull;
    return v ^ (v >> 31);
  };
  z = mix(mix(i ^ mix(z)) + s * 0x2545F4914F6CDD1Dull);
  return (z >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "Usage: %s <box_size> <num_spheres> [steps]\n", argv[0]);
    return 1;
  }
  const double box_size = std::atof(argv[1]);
  const size_t n = std::strtoull(argv[2], nullptr, 10);
  const int steps = argc > 3 ? std::atoi(argv[3]) : 1;
  // simulation parameters of NgpLcp.cpp:846-852
  const double viscosity = 0.001, dt = 5e-3, sphere_radius = 1.0, search_buffer = 3 * sphere_radius;
  const double max_allowable_overlap = 1e-5;
  const unsigned max_col_iterations = 10000;
  const double vf = (4.0 / 3.0 * M_PI * sphere_radius * sphere_radius * sphere_radius * n) / (box_size * box_size * box_size);
  std::printf("Initializing %zu spheres at a volume fraction of %g\n", n, vf);

  std::vector<double> pos(3 * n), rad(n, sphere_radius), mob(n, 1.0 / (6.0 * M_PI * sphere_radius * viscosity));
  for (size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) pos[3 * i + k] = u01(1234, i, k) * box_size;
  DeviceVector center(pos), radius(rad), mobility(mob), aabb(6 * n);

  mesh::GenNeighborLinks links;
  links.set_search_buffer(search_buffer).set_search_kind(MHIP_SEARCH_SPHERES).concretize();
  for (int step = 0; step < steps; ++step) {
    const auto t0 = std::chrono::steady_clock::now();
    check(mhip_compute_aabb_spheres(n, center.data(), radius.data(), aabb.data(), nullptr));
    std::printf("Generating neighbor pairs\n");
    const bool rebuilt = links.generate(n, aabb.data(), center.data(), radius.data());
    const size_t C = links.num_links();
    std::printf("Number of neighbor pairs: %zu%s\n", C, rebuilt ? "" : " (list reused)");
    auto pairs = links.links();
    std::printf("Computing signed separation distance and contact normal\n");
    DeviceVector sep(C), normal(3 * C);
    check(mhip_contact_spheres(C, pairs.data(), center.data(), radius.data(), nullptr, sep.data(), normal.data(),
                               nullptr));
    std::printf("Resolving initial collisions\n");
    ContactOperator op(C, n, pairs.data(), normal.data(), nullptr, nullptr, mobility.data(), nullptr, dt);
    DeviceVector lam(std::vector<double>(C, 0.0)), lam_tmp(C), g(C), g_tmp(C);
    mhip_solve_result res{};
    double max_speed = 0;
    check(mhip_scrap_bbpgd_solve_contact(op.handle(), sep.data(), max_allowable_overlap, max_col_iterations,
                                         lam.data(), lam_tmp.data(), g.data(), g_tmp.data(), &res, &max_speed,
                                         nullptr));
    const double* vel = nullptr;
    check(mhip_contact_op_body_velocity(op.handle(), &vel));
    check(mhip_integrate_euler(n, dt, vel, center.data(), nullptr, nullptr));  // Euler step (:898)
    check(mhip_stream_synchronize(nullptr));
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("Time to resolve collisions: %g seconds\n", secs);
    std::printf("Result: \n  Max abs projected sep: %g\n  Number of iterations: %u\n  Max displacement: %g\n",
                res.residual, res.num_iters, max_speed * dt);
    if (max_speed * dt > 2 * sphere_radius)
      std::printf("***WARNING*** The maximum displacement is larger than the search buffer. Collisions may be missed. "
                  "***WARNING***\n");
    if (!res.converged) std::printf("(iteration cap reached before the overlap tolerance)\n");
  }
  // the original's (commented-out) N^2 overlap check, on a sample: after the last step no pair overlaps by more than
  // the tolerance times a safety factor (the step is linearised)
  const auto p = center.download();
  size_t bad = 0;
  const size_t m = n < 2000 ? n : 2000;
  for (size_t a = 0; a < m; ++a)
    for (size_t b = a + 1; b < m; ++b) {
      const double dx = p[3 * a] - p[3 * b], dy = p[3 * a + 1] - p[3 * b + 1], dz = p[3 * a + 2] - p[3 * b + 2];
      if (std::sqrt(dx * dx + dy * dy + dz * dz) - 2.0 * sphere_radius < -0.05) ++bad;
    }
  std::printf("%s\n", bad ? "Overlap detected!" : "No overlap detected!");
  return bad ? 3 : 0;
}
