// Dev aid: semantics of the DPP controls the lane-per-entry kernel relies on (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL>
__device__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
__global__ void k(int* out) {
  int l = threadIdx.x;
  out[0 * 64 + l] = dpp<0x39>(l);        // quad_perm [1,2,3,0]
  out[1 * 64 + l] = dpp<0x4E>(l);        // quad_perm [2,3,0,1]
  out[2 * 64 + l] = dpp<0x93>(l);        // quad_perm [3,0,1,2]
  out[3 * 64 + l] = dpp<0x120 + 12>(l);  // row_ror:12
  out[4 * 64 + l] = dpp<0x120 + 8>(l);   // row_ror:8
  out[5 * 64 + l] = dpp<0x120 + 4>(l);   // row_ror:4
  out[6 * 64 + l] = dpp<0x150 + 7>(l);   // row_newbcast:7
  double x = l + 0.5;
  long long b = __builtin_bit_cast(long long, x);
  long long r = __builtin_amdgcn_update_dpp(0LL, b, 0x150 + 11, 0xF, 0xF, false);  // 64-bit row_newbcast:11
  out[7 * 64 + l] = (int)__builtin_bit_cast(double, r);
}
int main() {
  int* d; hipMalloc(&d, 8 * 64 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[8 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* nm[8] = {"quad[1,2,3,0]", "quad[2,3,0,1]", "quad[3,0,1,2]", "row_ror:12", "row_ror:8", "row_ror:4", "row_newbcast:7", "b64 newbcast:11"};
  for (int r = 0; r < 8; ++r) { printf("%-16s", nm[r]); for (int l = 0; l < 20; ++l) printf(" %2d", h[r * 64 + l]); printf("\n"); }
  return 0;
}
