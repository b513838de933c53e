#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
  __shared__ __attribute__((aligned(16))) short sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = i;
  __syncthreads();
  int lane = threadIdx.x;
  int q = (lane & 15) >> 2, p = lane & 3, grp = lane >> 4;
  __attribute__((address_space(3))) s4* ptr = (__attribute__((address_space(3))) s4*)(sm + q * 64 + grp * 16 + 4 * p);
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(ptr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short* d; hipMalloc(&d, 512); short h[256];
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
    int exp = e * 64 + (lane >> 4) * 16 + (lane & 15);
    if (h[lane * 4 + e] != exp) { if (bad < 8) printf("lane %d e %d got %d exp %d\n", lane, e, h[lane*4+e], exp); ++bad; }
  }
  printf("bad=%d\n", bad);
  return 0;
}
