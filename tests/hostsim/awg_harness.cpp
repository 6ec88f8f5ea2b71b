// Host run of the reverse sweep ekf_adjoint_wg_kernel<R, NE> (cdkf_adjoint_wg_kernels.h: the gradient of the marginal log-likelihood
// w.r.t. every parameter, one workgroup of 256 threads per trajectory) under the CPU sanitizers -- the library's instantiations or the
// translation unit launch_custom.hip generates for a drift given as source (force-included in front of this file, compiled with
// -DCDKF_HOST_SIM).  The forward sweep's four moment arrays come from tests/hostsim/wg_harness.cpp's run of the filter kernel; the
// argument struct, parameter block and geometry are the launcher's own (cdkf_debug_wg_args with smoother = 2).  Test infrastructure.
//   -DHS_REAL=double|float -DHS_NE=<8|16> -DCDKF_WG_STATIC_LDS=<bytes>
//   in : int64 head[16] = {N, threads, n_args_bytes, n_blob, n_t, n_y, n_m, n_P, n_grad, n_grad_model, ws_stride, cap, n_u, 0, 0, 0},
//        WgArgs bytes, blob, t, y, fm[n_m], fP[n_P], pm[n_m], pP[n_P] (the forward sweep's outputs, [T,N,...]), u[n_u]
//   out: ll[N], grad[n_grad], grad_model[n_grad_model], status[N]
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef HS_REAL HR;

template <typename T>
static T* rd(FILE* f, long n) {
  T* p = (T*)malloc((n > 0 ? n : 1) * sizeof(T));
  if (n > 0 && fread(p, sizeof(T), n, f) != (size_t)n) {
    fprintf(stderr, "awg_harness: short input\n");
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
    fprintf(stderr, "awg_harness: WgArgs is %zu bytes here, %lld in the library\n", sizeof(cdkf::WgArgs<HR>), (long long)head[2]);
    return 3;
  }
  cdkf::WgArgs<HR> a;
  {
    unsigned char* raw = rd<unsigned char>(f, head[2]);
    memcpy(&a, raw, sizeof(a));
    free(raw);
  }
  const long N = head[0];
  HR* blob = rd<HR>(f, head[3]);
  HR* t = rd<HR>(f, head[4]);
  HR* y = rd<HR>(f, head[5]);
  HR *fm = rd<HR>(f, head[6]), *fP = rd<HR>(f, head[7]), *pm = rd<HR>(f, head[6]), *pP = rd<HR>(f, head[7]);
  HR* u = head[12] > 0 ? rd<HR>(f, head[12]) : nullptr;
  fclose(f);
  // the kernel ACCUMULATES into the gradient arrays (the launcher zeroes them); scratch and ll need no initial value
  HR* grad = (HR*)calloc(head[8] > 0 ? head[8] : 1, sizeof(HR));
  HR* gm = head[9] > 0 ? (HR*)calloc(head[9], sizeof(HR)) : nullptr;
  HR* ws = (HR*)malloc((size_t)N * head[10] * sizeof(HR));
  HR* ll = (HR*)calloc(N, sizeof(HR));
  int* status = (int*)calloc(N, sizeof(int));
  a.par = blob;
  a.t = t;
  a.y = y;
  a.ll = ll;
  a.fm = fm;
  a.fP = fP;
  a.pm = pm;
  a.pP = pP;
  a.status = status;
  a.u = a.du > 0 ? u : nullptr;
  const long ws_stride = head[10];
  const int cap = (int)head[11];
  hostsim::launch((unsigned)N, (unsigned)head[1], [&] { cdkf::ekf_adjoint_wg_kernel<HR, HS_NE>(a, grad, gm, ws, ws_stride, cap); });
  FILE* g = fopen(argv[2], "wb");
  if (!g) return 2;
  auto wr = [&](const void* p, long n, size_t sz) {
    if (n > 0 && fwrite(p, sz, n, g) != (size_t)n) exit(2);
  };
  wr(ll, N, sizeof(HR));
  wr(grad, head[8], sizeof(HR));
  wr(gm, head[9], sizeof(HR));
  wr(status, N, sizeof(int));
  fclose(g);
  return 0;
}
