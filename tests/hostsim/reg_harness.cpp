// Host run of a run-time generated register-resident kernel (launch_custom.hip: generate_source -> `cdkf_custom_kernel`) under the CPU
// sanitizers: the generated translation unit is compiled for x86-64 with -DCDKF_HOST_SIM (cd_dynamax_amd/csrc/hostsim/cdkf_hostsim.h)
// and this file, which `-include`s it, calls the kernel for every (block, lane) of the launch in sequence -- the register-resident
// sweeps have no cross-lane traffic.  Test infrastructure (tests/test_hostsim.py); input / output: flat binary files.
//   in : int64 head[16] = {blocks, n_par, n_ip, n_t, n_y, n_ll, n_o1, n_o2, n_o3, n_o4, n_status, n_sm, n_sP, n_u, preload, 0},
//        par[n_par] (R), ip[n_ip] (int64), t[n_t] (R), y[n_y] (R), u[n_u] (R: the inputs, absent when n_u = 0)
//        [+ o1[n_o1], o2[n_o2] when preload = 1: the filtered moments the backward sweep of the smoother reads]
//   out: ll, o1 .. o4, status (int32), sm, sP, in that order (absent arrays have length 0 and are passed as null)
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <typename T>
static T* rd(FILE* f, long n) {
  T* p = (T*)malloc((n > 0 ? n : 1) * sizeof(T));
  if (n > 0 && fread(p, sizeof(T), n, f) != (size_t)n) {
    fprintf(stderr, "reg_harness: short input\n");
    exit(2);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int64_t* head = rd<int64_t>(f, 16);
  R* par = rd<R>(f, head[1]);
  long* ip = (long*)rd<int64_t>(f, head[2]);
  R* t = rd<R>(f, head[3]);
  R* y = rd<R>(f, head[4]);
  R* u = head[13] > 0 ? rd<R>(f, head[13]) : nullptr;
  R *pre1 = nullptr, *pre2 = nullptr;
  if (head[14]) {
    pre1 = rd<R>(f, head[6]);
    pre2 = rd<R>(f, head[7]);
  }
  fclose(f);
  // malloc, not calloc: what the kernel leaves unwritten stays poisoned for MemorySanitizer and is reported when it is written out
  auto out = [](long n) { return n > 0 ? (R*)malloc(n * sizeof(R)) : (R*)nullptr; };
  R *ll = out(head[5]), *o1 = pre1 ? pre1 : out(head[6]), *o2 = pre2 ? pre2 : out(head[7]), *o3 = out(head[8]), *o4 = out(head[9]);
  int* status = head[10] > 0 ? (int*)(head[14] ? calloc(head[10], sizeof(int)) : malloc(head[10] * sizeof(int))) : nullptr;
  R *sm = out(head[11]), *sP = out(head[12]);
  hostsim::launch_serial((unsigned)head[0], 64, [&] { cdkf_custom_kernel(par, ip, t, y, ll, o1, o2, o3, o4, status, sm, sP, u); });
  FILE* g = fopen(argv[2], "wb");
  if (!g) return 2;
  auto wr = [&](const void* p, long n, size_t sz) {
    if (n > 0 && fwrite(p, sz, n, g) != (size_t)n) exit(2);
  };
  if (!head[14]) wr(ll, head[5], sizeof(R));  // (the backward sweep writes no log-likelihood)
  wr(o1, head[6], sizeof(R));
  wr(o2, head[7], sizeof(R));
  wr(o3, head[8], sizeof(R));
  wr(o4, head[9], sizeof(R));
  wr(status, head[10], sizeof(int));
  wr(sm, head[11], sizeof(R));
  wr(sP, head[12], sizeof(R));
  fclose(g);
  return 0;
}
