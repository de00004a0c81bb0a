// prints the lane -> element mapping of ds_read_b64_tr_b16 (tile[r][c] = r * 100 + c, 64-column rows)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short *out) {
  __shared__ short tile[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) tile[i] = (short)((i / 64) * 100 + i % 64);
  __syncthreads();
  // lane 4q+p of a 16-lane group g supplies the address of row (4g + q), columns 4p..4p+3
  const int l = threadIdx.x, g = l / 16, q = (l % 16) / 4, p = l % 4;
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s *)(tile + (4 * g + q) * 64 + p * 4));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = r[e];
}
int main() {
  short *d, h[256];
  hipMalloc(&d, 512);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
