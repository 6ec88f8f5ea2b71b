// Host run of the emission-moments kernel generated around a model's emission statements (launch_custom.hip: kEmissionMomentsKernel,
// dumped by cdkf_custom_emission_moments_compile under CDKF_CUSTOM_DUMP) under the CPU sanitizers: compiled for x86-64 with
// -DCDKF_HOST_SIM and force-included in front of this file; a lane per (m, P) row, no cross-lane traffic.  Test infrastructure.
//   in : int64 head[4] = {rows, ukf, with_cov, n_par}, R w[4] = {c, wm0, wc0, wi}, par, t [rows], u [rows, DU] (DU > 0), mu [rows, D],
//        P [rows, D, D] (with_cov)
//   out: ym [rows, M], yc [rows, M, M] (with_cov)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

template <typename T>
static T* rd(FILE* f, long n) {
  T* p = (T*)malloc((n > 0 ? n : 1) * sizeof(T));
  if (n > 0 && fread(p, sizeof(T), n, f) != (size_t)n) {
    fprintf(stderr, "em_harness: short input\n");
    exit(2);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  constexpr int D = cdkf::UtModel::D, M = cdkf::UtModel::M, DU = cdkf::UtModel::DU;
  int64_t* head = rd<int64_t>(f, 4);
  const long rows = head[0];
  R* w = rd<R>(f, 4);
  EmArgs a;
  memset(&a, 0, sizeof(a));
  a.par = rd<R>(f, head[3]);
  a.t = rd<R>(f, rows);
  a.u = DU > 0 ? rd<R>(f, rows * DU) : nullptr;
  a.mu = rd<R>(f, rows * D);
  a.P = head[2] ? rd<R>(f, rows * D * D) : nullptr;
  fclose(f);
  // malloc, not calloc: what the kernel leaves unwritten stays poisoned for MemorySanitizer
  a.ym = (R*)malloc(rows * M * sizeof(R));
  a.yc = head[2] ? (R*)malloc(rows * M * M * sizeof(R)) : nullptr;
  a.rows = rows;
  a.c = w[0]; a.wm0 = w[1]; a.wc0 = w[2]; a.wi = w[3];
  a.ukf = (int)head[1];
  hostsim::launch_serial((unsigned)((rows + 63) / 64), 64, [&] { cdkf_emission_moments_kernel(a); });
  FILE* g = fopen(argv[2], "wb");
  if (!g) return 2;
  if (fwrite(a.ym, sizeof(R), rows * M, g) != (size_t)(rows * M)) return 2;
  if (a.yc && fwrite(a.yc, sizeof(R), rows * M * M, g) != (size_t)(rows * M * M)) return 2;
  fclose(g);
  return 0;
}
