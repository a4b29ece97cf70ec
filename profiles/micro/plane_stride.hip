// Microbenchmark: does the 16-plane access pattern of the planar image kernels lose bandwidth because all
// planes of an [N,C,H,W] tensor with a power-of-two plane stride (2048*2048*4 B = 16 MiB) alias onto the
// same HBM channel/bank?  Two probes against the baseline (stride = HW, every plane at the same offset):
//   pad  : plane stride HW + pad elements (a layout change)
//   skew : plane p of workgroup b works on chunk (b + p*skew) mod nchunks (same layout, other schedule)
//   hipcc --offload-arch=gfx950 -O3 -o plane_stride plane_stride.hip && ./plane_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

template <bool STORE>
__global__ __launch_bounds__(256) void k(float* buf, long HW, long PS, int skew, int nchunks, float* sink) {
  const long n = blockIdx.y;
  float* o = buf + n * 16 * PS;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 v = make_float4(1.f, 2.f, 3.f, float(threadIdx.x));
#pragma unroll
  for (int p = 0; p < 16; ++p) {
    int chunk = blockIdx.x + p * skew;
    chunk = chunk % nchunks;
    float4* q = reinterpret_cast<float4*>(o + long(p) * PS + (long(chunk) * 256 + threadIdx.x) * 4);
    if (STORE) *q = v;
    else { const float4 t = *q; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
  }
  if (!STORE && acc.x + acc.y + acc.z + acc.w == 12345.f) *sink = 1.f;
}

template <bool STORE>
void run(float* buf, long N, long HW, long pad, int skew) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int nchunks = int(HW / 1024);
  const dim3 grid((unsigned)nchunks, (unsigned)N);
  const long PS = HW + pad;
  hipLaunchKernelGGL((k<STORE>), grid, dim3(256), 0, 0, buf, HW, PS, skew, nchunks, buf);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<STORE>), grid, dim3(256), 0, 0, buf, HW, PS, skew, nchunks, buf);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 10;
  printf("%s pad=%7ld el skew=%5d chunks : %.3f ms  %.2f TB/s\n", STORE ? "store" : "load ", pad, skew, ms,
         double(N) * 16 * HW * 4 / ms * 1e-9);
}

int main() {
  const long N = 8, HW = 2048L * 2048L;
  float* buf;
  CK(hipMalloc(&buf, (N * 16 * (HW + 65536) + 1024) * 4));
  CK(hipMemset(buf, 0, (N * 16 * (HW + 65536) + 1024) * 4));
  const long pads[] = {0, 64, 256, 1024, 1088, 4096, 4160, 16384 + 1024, 65536 - 1024};
  for (long pad : pads) run<true>(buf, N, HW, pad, 0);
  const int skews[] = {1, 2, 3, 17, 64, 65, 257, 1031};
  for (int s : skews) run<true>(buf, N, HW, 0, s);
  for (long pad : pads) run<false>(buf, N, HW, pad, 0);
  for (int s : skews) run<false>(buf, N, HW, 0, s);
  return 0;
}
