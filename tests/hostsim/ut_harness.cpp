// Host run of the tangent sweep of the literal unscented recursion (cdkf_ukf_tangent_kernels.h; the translation unit launch_custom.hip
// generates around it: CDKF_CUSTOM_DUMP) under the CPU sanitizers: compiled for x86-64 with -DCDKF_HOST_SIM and force-included in front of
// this file, every (block, lane) of the launch called in sequence -- a lane per (trajectory, leaf entry), no cross-lane traffic.  The
// argument struct and the parameter block are the launcher's own (cdkf_debug_ukf_tangent_args).  Test infrastructure (tests/test_hostsim.py).
//   in : int64 head[8] = {blocks, args bytes, n_par, n_t, n_y, n_u, n_grad, n_grad_model}, UtArgs bytes, par (R), t (R), y (R), u (R)
//   out: ll [N], grad, grad_model, status [N] (int32) [, value mode: filtered means, covs, predicted means, covs]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

template <typename T>
static T* rd(FILE* f, long n) {
  T* p = (T*)malloc((n > 0 ? n : 1) * sizeof(T));
  if (n > 0 && fread(p, sizeof(T), n, f) != (size_t)n) {
    fprintf(stderr, "ut_harness: short input\n");
    exit(2);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int64_t* head = rd<int64_t>(f, 8);
  if (head[1] != (int64_t)sizeof(cdkf::UtArgs<R>)) {
    fprintf(stderr, "ut_harness: UtArgs is %zu bytes here, %lld in the library\n", sizeof(cdkf::UtArgs<R>), (long long)head[1]);
    return 3;
  }
  cdkf::UtArgs<R> a;
  {
    unsigned char* raw = rd<unsigned char>(f, head[1]);
    memcpy(&a, raw, sizeof(a));
    free(raw);
  }
  R* par = rd<R>(f, head[2]);
  R* t = rd<R>(f, head[3]);
  R* y = rd<R>(f, head[4]);
  R* u = head[5] > 0 ? rd<R>(f, head[5]) : nullptr;
  fclose(f);
  // malloc, not calloc: what the kernel leaves unwritten stays poisoned for MemorySanitizer and is reported when it is written out
  R* ll = (R*)malloc(a.N * sizeof(R));
  R* grad = (R*)malloc((head[6] > 0 ? head[6] : 1) * sizeof(R));
  R* gm = head[7] > 0 ? (R*)malloc(head[7] * sizeof(R)) : nullptr;
  int* status = (int*)malloc(a.N * sizeof(int));
  a.par = par; a.t = t; a.y = y; a.u = u; a.ll = ll; a.grad = grad; a.grad_model = gm; a.status = status;
  // value mode (UtArgs::value_only: the filter through the same kernel): the four moment arrays, [N, T, D] and [N, T, D * D] as the
  // strides of the argument block address them (contiguous: the harness is run with the default layout)
  const long nm = a.N * a.T * cdkf::UtModel::D, nP = nm * cdkf::UtModel::D;
  R* mom[4] = {nullptr, nullptr, nullptr, nullptr};
  if (a.value_only) {
    for (int k = 0; k < 4; ++k) mom[k] = (R*)malloc(((k & 1) ? nP : nm) * sizeof(R));
    a.fm = mom[0]; a.fc = mom[1]; a.pm = mom[2]; a.pc = mom[3];
  }
  hostsim::launch_serial((unsigned)head[0], 64, [&] { cdkf_ukf_tangent_kernel(a); });
  FILE* g = fopen(argv[2], "wb");
  if (!g) return 2;
  auto wr = [&](const void* p, long n, size_t sz) {
    if (n > 0 && fwrite(p, sz, n, g) != (size_t)n) exit(2);
  };
  wr(ll, a.N, sizeof(R));
  wr(grad, head[6], sizeof(R));
  wr(gm, head[7], sizeof(R));
  wr(status, a.N, sizeof(int));
  if (a.value_only)
    for (int k = 0; k < 4; ++k) wr(mom[k], (k & 1) ? nP : nm, sizeof(R));
  fclose(g);
  return 0;
}
