// example_register.cpp — the C ABI from compiled code, nothing else: reads a correspondence file (one `x y z x' y' z'`
// per line), calls sc_register, prints the result.  What a C++ registration pipeline would add.
//
//   g++ -O2 -std=c++17 -I include integration/example_register.cpp -L sac-cot_amd -lsaccot \
//       -Wl,-rpath,$PWD/sac-cot_amd -o integration/example_register
//   integration/example_register corr.txt <tau> [T] [--refine] [--devices 0,1,2,3 | --loopback N]
// --devices: the native multi-device entry (sc_create_multi / sc_register_multi: RCCL inside the library); one device
// is exactly sc_register.  --loopback N: N ranks on device 0 without RCCL (the test hook, for one-GPU boxes).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "saccot.h"

int main(int argc, char** argv) {
  if (argc < 3) {
    std::fprintf(stderr, "usage: %s corr.txt tau [T] [--refine] [--devices 0,1,... | --loopback N]\n", argv[0]);
    return 2;
  }
  const float tau = std::strtof(argv[2], nullptr);
  unsigned T = 50000;
  bool refine = false;
  std::vector<int> devices;
  int loopback = 0;
  for (int a = 3; a < argc; a++) {
    if (!std::strcmp(argv[a], "--refine")) refine = true;
    else if (!std::strcmp(argv[a], "--devices") && a + 1 < argc) {
      for (char* tok = std::strtok(argv[++a], ","); tok; tok = std::strtok(nullptr, ",")) devices.push_back(std::atoi(tok));
    } else if (!std::strcmp(argv[a], "--loopback") && a + 1 < argc) loopback = std::atoi(argv[++a]);
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

  const bool multi = !devices.empty() || loopback > 0;
  sc_ctx* ctx = nullptr;
  sc_multi* mg = nullptr;
  int rc;
  if (multi) {
    rc = loopback > 0 ? sc_create_multi_loopback(0, loopback, &mg) : sc_create_multi(devices.data(), (int)devices.size(), &mg);
    if (rc != SC_OK) { std::fprintf(stderr, "sc_create_multi: %s\n", sc_strerror(rc)); return 1; }
  } else {
    rc = sc_create(0, &ctx);  // no GPU -> SC_EHIP: there is no CPU fallback
    if (rc != SC_OK) { std::fprintf(stderr, "sc_create: %s\n", sc_strerror(rc)); return 1; }
  }
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
  rc = multi ? sc_register_multi(mg, src.data(), tgt.data(), n, &p, R, t, mask.data(), &st)
             : sc_register(ctx, src.data(), tgt.data(), n, &p, R, t, mask.data(), &st);
  if (rc != SC_OK && rc != SC_ENOHYP) {
    std::fprintf(stderr, "sc_register: %s (%s)\n", sc_strerror(rc), multi ? sc_multi_last_error(mg) : sc_last_error(ctx));
    sc_destroy(ctx); sc_destroy_multi(mg);
    return 1;
  }
  size_t inl = 0;
  for (uint8_t m : mask) inl += m;
  std::printf("abi %d.%d n %lld status %d edges %llu triangles_kept %u best_rank %u inliers %zu\n", sc_version() >> 16,
              sc_version() & 0xFFFF, (long long)n, rc, (unsigned long long)st.edges, st.tri_kept, st.best_rank, inl);
  std::printf("R %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g %.9g\nt %.9g %.9g %.9g\n", R[0], R[1], R[2], R[3], R[4], R[5],
              R[6], R[7], R[8], t[0], t[1], t[2]);
  sc_destroy(ctx);
  sc_destroy_multi(mg);
  return rc == SC_OK ? 0 : 3;
}
