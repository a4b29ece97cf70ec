// Microbenchmark: HBM write bandwidth of the "one 16-byte store per lane into each of P channel planes"
// pattern of the planar image writers (interpolate forward writes P = C = 16 planes), against a single
// linear stream, and against variants that keep a workgroup on one plane for longer.
//   hipcc --offload-arch=gfx950 -O3 -o store_planes store_planes.hip && ./store_planes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

// grid.x = HW / (256*4*TILE), grid.y = N.  Every lane owns 4 consecutive pixels; a workgroup walks TILE
// consecutive 1024-pixel chunks.  ORDER 0: for chunk { for plane { store } }   (planes innermost)
//                                  ORDER 1: for plane { for chunk { store } }   (a plane at a time)
template <int P, int TILE, int ORDER>
__global__ __launch_bounds__(256) void k(float* out, long HW) {
  const long n = blockIdx.y;
  const long base = (long(blockIdx.x) * TILE * 256 + threadIdx.x) * 4;
  float* o = out + n * P * HW;
  const float4 v = make_float4(1.f, 2.f, 3.f, float(threadIdx.x));
  if (ORDER == 0) {
    for (int t = 0; t < TILE; ++t)
#pragma unroll
      for (int p = 0; p < P; ++p) *reinterpret_cast<float4*>(o + long(p) * HW + base + long(t) * 1024) = v;
  } else {
#pragma unroll 1
    for (int p = 0; p < P; ++p)
      for (int t = 0; t < TILE; ++t) *reinterpret_cast<float4*>(o + long(p) * HW + base + long(t) * 1024) = v;
  }
}

template <int P, int TILE, int ORDER>
void run(float* buf, long N, long HW, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const dim3 grid((unsigned)(HW / (1024 * TILE)), (unsigned)(N * 16 / P));
  hipLaunchKernelGGL((k<P, TILE, ORDER>), grid, dim3(256), 0, 0, buf, HW);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<P, TILE, ORDER>), grid, dim3(256), 0, 0, buf, HW);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 10;
  printf("%-34s %.3f ms  %.2f TB/s\n", name, ms, double(N) * 16 * HW * 4 / ms * 1e-9);
}

int main() {
  const long N = 8, HW = 2048L * 2048L;
  float* buf;
  CK(hipMalloc(&buf, N * 16 * HW * 4));
  run<1, 1, 0>(buf, N, HW, "P=1  (linear)");
  run<4, 1, 0>(buf, N, HW, "P=4  planes innermost");
  run<16, 1, 0>(buf, N, HW, "P=16 planes innermost (today)");
  run<16, 4, 0>(buf, N, HW, "P=16 TILE=4 planes innermost");
  run<16, 4, 1>(buf, N, HW, "P=16 TILE=4 plane at a time");
  run<16, 16, 1>(buf, N, HW, "P=16 TILE=16 plane at a time");
  run<16, 64, 1>(buf, N, HW, "P=16 TILE=64 plane at a time");
  return 0;
}
