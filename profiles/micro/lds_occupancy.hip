// How many 256-thread workgroups of a kernel with a given amount of LDS does a CU of this device hold?  (the allocation
// granule decides whether 5 x 32 144 B fit the 160 KB)   hipcc --offload-arch=gfx950 -O2 -o lds_occupancy lds_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void probe(float* out) {
  extern __shared__ float s[];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = s[255 - threadIdx.x];
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("%s: sharedMemPerMultiprocessor %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.name, p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor);
  int prev = -1;
  for (int bytes = 20 * 1024; bytes <= 64 * 1024; bytes += 128) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, probe, 256, bytes) != hipSuccess) { printf("query failed at %d\n", bytes); break; }
    if (n != prev) printf("from %6d B of LDS per workgroup: %d workgroups per CU\n", bytes, n);
    prev = n;
  }
  return 0;
}
