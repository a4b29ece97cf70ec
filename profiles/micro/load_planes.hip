// Microbenchmark: HBM read bandwidth when every lane reads from P channel planes (4 B or 16 B per lane and
// plane), the access pattern of the planar image readers (interpolate backward: 16 grad_out planes at
// 4 B/lane; edge_dots: 2 x 16 planes at 16 B/lane), against a single linear stream.
//   hipcc --offload-arch=gfx950 -O3 -o load_planes load_planes.hip && ./load_planes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

template <int P, int VEC>
__global__ __launch_bounds__(256) void k(const float* in, float* sink, long HW) {
  const long n = blockIdx.y;
  const long base = (long(blockIdx.x) * 256 + threadIdx.x) * VEC;
  const float* p0 = in + n * P * HW + base;
  float acc = 0.f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    if (VEC == 4) {
      const float4 v = *reinterpret_cast<const float4*>(p0 + long(p) * HW);
      acc += v.x + v.y + v.z + v.w;
    } else {
      acc += p0[long(p) * HW];
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

template <int P, int VEC>
void run(const float* buf, float* sink, long N, long HW, const char* name) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const dim3 grid((unsigned)(HW / (256 * VEC)), (unsigned)(N * 16 / P));
  hipLaunchKernelGGL((k<P, VEC>), grid, dim3(256), 0, 0, buf, sink, HW);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<P, VEC>), grid, dim3(256), 0, 0, buf, sink, HW);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 10;
  printf("%-34s %.3f ms  %.2f TB/s\n", name, ms, double(N) * 16 * HW * 4 / ms * 1e-9);
}

int main() {
  const long N = 8, HW = 2048L * 2048L;
  float *buf, *sink;
  CK(hipMalloc(&buf, N * 16 * HW * 4));
  CK(hipMalloc(&sink, 16));
  CK(hipMemset(buf, 0, N * 16 * HW * 4));
  run<1, 4>(buf, sink, N, HW, "P=1  16 B/lane (linear)");
  run<16, 4>(buf, sink, N, HW, "P=16 16 B/lane");
  run<1, 1>(buf, sink, N, HW, "P=1   4 B/lane (linear)");
  run<4, 1>(buf, sink, N, HW, "P=4   4 B/lane");
  run<16, 1>(buf, sink, N, HW, "P=16  4 B/lane");
  return 0;
}
