// example_register.cpp — the C ABI from compiled code, nothing else: reads a correspondence file (one `x y z x' y' z'`
// per line), calls sc_register, prints the result.  What a C++ registration pipeline would add.
//
//   g++ -O2 -std=c++17 -I include integration/example_register.cpp -L sac-cot_amd -lsaccot \
//       -Wl,-rpath,$PWD/sac-cot_amd -o integration/example_register
//   integration/example_register corr.txt <tau> [T] [--refine]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "saccot.h"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s corr.txt tau [T] [--refine]\n", argv[0]);
    return 2;
  }
  const float tau = std::strtof(argv[2], nullptr);
  unsigned T = 50000;
  bool refine = false;
  for (int a = 3; a < argc; a++) {
    if (!std::strcmp(argv[a], "--refine")) refine = true;
    else T = (unsigned)std::strtoul(argv[a], nullptr, 10);
  }
  std::FILE* f = std::fopen(argv[1], "r");
  if (!f) { std::perror(argv[1]); return 2; }
  std::vector<float> src, tgt;
  char line[512];
  while (std::fgets(line, sizeof line, f)) {
    float v[6];
    if (line[0] == '#') continue;
    if (std::sscanf(line, "%f %f %f %f %f %f", v, v + 1, v + 2, v + 3, v + 4, v + 5) != 6) continue;
    src.insert(src.end(), v, v + 3);
    tgt.insert(tgt.end(), v + 3, v + 6);
  }
  std::fclose(f);
  const int64_t n = (int64_t)src.size() / 3;

  sc_ctx* ctx = nullptr;
  int rc = sc_create(0, &ctx);  // no GPU -> SC_EHIP: there is no CPU fallback
  if (rc != SC_OK) { std::fprintf(stderr, "sc_create: %s\n", sc_strerror(rc)); return 1; }
  sc_params p;
  sc_default_params(&p);
  p.sigma = tau; p.tau = tau; p.min_len = tau; p.t_cmp = 0.9f;
  p.max_triangles = T;
  if (refine) p.flags |= SC_FLAG_REFINE;
  std::vector<uint8_t> mask((size_t)n);
  float R[9], t[3];
  sc_stats st;
  std::memset(&st, 0, sizeof st);
  st.size = sizeof st;
  rc = sc_register(ctx, src.data(), tgt.data(), n, &p, R, t, mask.data(), &st);
  if (rc != SC_OK && rc != SC_ENOHYP) {
    std::fprintf(stderr, "sc_register: %s (%s)\n", sc_strerror(rc), sc_last_error(ctx));
    sc_destroy(ctx);
    return 1;
  }
  size_t inl = 0;
  for (uint8_t m : mask) inl += m;
  std::printf("abi %d.%d n %lld status %d edges %llu triangles_kept %u best_rank %u inliers %zu\n", sc_version() >> 16,
              sc_version() & 0xFFFF, (long long)n, rc, (unsigned long long)st.edges, st.tri_kept, st.best_rank, inl);
  std::printf("R %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\nt %.9g %.9g %.9g\n", R[0], R[1], R[2], R[3], R[4], R[5],
              R[6], R[7], R[8], t[0], t[1], t[2]);
  sc_destroy(ctx);
  return rc == SC_OK ? 0 : 3;
}
