// Host run of ONE instantiation of the workgroup-per-trajectory kernels (cdkf_wg2_kernels.h) under the CPU sanitizers: the kernel's
// translation unit -- the library's headers, or the source launch_custom.hip generates for a drift given as C source -- is compiled
// for x86-64 with -DCDKF_HOST_SIM (cd_dynamax_amd/csrc/hostsim/cdkf_hostsim.h) and force-included in front of this file; every GPU
// thread of a workgroup is an OS thread, __syncthreads() a barrier ThreadSanitizer understands, the LDS block a static array with
// AddressSanitizer's red zones around it.  The argument struct and the parameter block are the launcher's own (cdkf_debug_wg_args).
// Test infrastructure (tests/test_hostsim.py).  Build-time selection of the instantiation:
//   -DHS_REAL=double|float -DHS_EPT=<entries per thread> -DHS_UKF=<0|1> -DHS_KIND=<drift kind or -1> -DHS_SMOOTHER=<0|1>
//   -DCDKF_WG_STATIC_LDS=<bytes>   (the carve-up's size, as for the run-time compiled variants)
//   in : int64 head[16] = {N, threads, n_args_bytes, n_blob, n_t, n_y, n_ll, n_fm, n_fP, n_pm, n_pP, n_status, n_sm, n_sP, n_u, 0},
//        WgArgs bytes, blob (R), t (R), y (R)  [+ fm, fP (R) when HS_SMOOTHER: the filtered moments the backward sweep reads]
//        [+ u (R), n_u > 0: the inputs a drift given as source reads]
//   out: ll, fm, fP, pm, pP, status (int32), sm, sP
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef HS_REAL HR;

template <typename T>
static T* rd(FILE* f, long n) {
  T* p = (T*)malloc((n > 0 ? n : 1) * sizeof(T));
  if (n > 0 && fread(p, sizeof(T), n, f) != (size_t)n) {
    fprintf(stderr, "wg_harness: short input\n");
    exit(2);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc != 3) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  int64_t* head = rd<int64_t>(f, 16);
  if (head[2] != (int64_t)sizeof(cdkf::WgArgs<HR>)) {
    fprintf(stderr, "wg_harness: WgArgs is %zu bytes here, %lld in the library\n", sizeof(cdkf::WgArgs<HR>), (long long)head[2]);
    return 3;
  }
  cdkf::WgArgs<HR> a;
  {
    unsigned char* raw = rd<unsigned char>(f, head[2]);
    memcpy(&a, raw, sizeof(a));
    free(raw);
  }
  HR* blob = rd<HR>(f, head[3]);
  HR* t = rd<HR>(f, head[4]);
  HR* y = rd<HR>(f, head[5]);
  auto out = [](long n) { return n > 0 ? (HR*)malloc(n * sizeof(HR)) : (HR*)nullptr; };
  HR *ll = out(head[6]), *fm = out(head[7]), *fP = out(head[8]), *pm = out(head[9]), *pP = out(head[10]);
  int* status = head[11] > 0 ? (int*)calloc(head[11], sizeof(int)) : nullptr;
  HR *sm = out(head[12]), *sP = out(head[13]);
#if HS_SMOOTHER
  if (fread(fm, sizeof(HR), head[7], f) != (size_t)head[7] || fread(fP, sizeof(HR), head[8], f) != (size_t)head[8]) return 2;
#endif
  HR* u = head[14] > 0 ? rd<HR>(f, head[14]) : nullptr;
  fclose(f);
  a.par = blob;
  a.t = t;
  a.y = y;
  a.ll = ll;
  a.fm = fm;
  a.fP = fP;
  a.pm = pm;
  a.pP = pP;
  a.sm = sm;
  a.sP = sP;
  a.status = status;
  a.u = a.du > 0 ? u : nullptr;   // (strides and du: the launcher's own, cdkf_debug_wg_args)
  hostsim::launch((unsigned)head[0], (unsigned)head[1], [&] {
#if HS_SMOOTHER
    cdkf::ekf_smoother_wg_kernel<HR, HS_EPT>(a);
#else
    cdkf::ekf_filter_wg_kernel<HR, HS_EPT, (HS_UKF != 0), HS_KIND>(a);
#endif
  });
  FILE* g = fopen(argv[2], "wb");
  if (!g) return 2;
  auto wr = [&](const void* p, long n, size_t sz) {
    if (n > 0 && fwrite(p, sz, n, g) != (size_t)n) exit(2);
  };
#if HS_SMOOTHER
  wr(sm, head[12], sizeof(HR));
  wr(sP, head[13], sizeof(HR));
  wr(status, head[11], sizeof(int));
#else
  wr(ll, head[6], sizeof(HR));
  wr(fm, head[7], sizeof(HR));
  wr(fP, head[8], sizeof(HR));
  wr(pm, head[9], sizeof(HR));
  wr(pP, head[10], sizeof(HR));
  wr(status, head[11], sizeof(int));
#endif
  fclose(g);
  return 0;
}
