// Microbenchmark, follow-up of plane_stride.hip: the layout stays [N,16,H,W] with the 16 MiB plane stride;
// a workgroup owns T consecutive 4 KB chunks (1024 px) of all 16 planes and issues its 16*T chunk-sized
// accesses in different ORDERS.  Which orders recover the bandwidth that plane_stride's global skew does?
//   O0: for t { for p { (p, t) } }                      planes innermost (what the kernels do today, T=1)
//   O1: for p { for t { (p, t) } }                      a plane at a time
//   O2: for g<4 { for t { for p in 4g..4g+3 { (p, (t+g)%T) } } }   plane groups, chunk rotated per group
//   O3: for t { for p { (p, (t+p)%T) } }                planes innermost, chunk rotated per plane
//   O4: for t { for g<4 { for p in 4g..4g+3 { (p, (t+g)%T) } } }   like O2, groups innermost
//   hipcc --offload-arch=gfx950 -O3 -o plane_order plane_order.hip && ./plane_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

template <bool STORE>
__device__ __forceinline__ void touch(float* o, long HW, int p, long chunk, float4& acc) {
  float4* q = reinterpret_cast<float4*>(o + long(p) * HW + (chunk * 256 + threadIdx.x) * 4);
  if (STORE) *q = make_float4(1.f, 2.f, 3.f, float(threadIdx.x));
  else { const float4 t = *q; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
}

template <bool STORE, int T, int ORDER>
__global__ __launch_bounds__(256) void k(float* buf, long HW, float* sink) {
  float* o = buf + long(blockIdx.y) * 16 * HW;
  const long c0 = long(blockIdx.x) * T;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ORDER == 0) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int p = 0; p < 16; ++p) touch<STORE>(o, HW, p, c0 + t, acc);
  } else if (ORDER == 1) {
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int t = 0; t < T; ++t) touch<STORE>(o, HW, p, c0 + t, acc);
  } else if (ORDER == 2) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int t = 0; t < T; ++t)
#pragma unroll
        for (int p = 4 * g; p < 4 * g + 4; ++p) touch<STORE>(o, HW, p, c0 + (t + g) % T, acc);
  } else if (ORDER == 3) {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int p = 0; p < 16; ++p) touch<STORE>(o, HW, p, c0 + (t + p) % T, acc);
  } else {
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int p = 4 * g; p < 4 * g + 4; ++p) touch<STORE>(o, HW, p, c0 + (t + g) % T, acc);
  }
  if (!STORE && acc.x + acc.y + acc.z + acc.w == 12345.f) *sink = 1.f;
}

template <bool STORE, int T, int ORDER>
void run(float* buf, long N, long HW) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const dim3 grid((unsigned)(HW / 1024 / T), (unsigned)N);
  hipLaunchKernelGGL((k<STORE, T, ORDER>), grid, dim3(256), 0, 0, buf, HW, buf);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<STORE, T, ORDER>), grid, dim3(256), 0, 0, buf, HW, buf);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 10;
  printf("%s T=%2d O%d : %.3f ms  %.2f TB/s\n", STORE ? "store" : "load ", T, ORDER, ms, double(N) * 16 * HW * 4 / ms * 1e-9);
}

template <bool STORE>
void all(float* buf, long N, long HW) {
  run<STORE, 1, 0>(buf, N, HW);
  run<STORE, 2, 0>(buf, N, HW); run<STORE, 2, 1>(buf, N, HW); run<STORE, 2, 3>(buf, N, HW);
  run<STORE, 4, 0>(buf, N, HW); run<STORE, 4, 1>(buf, N, HW); run<STORE, 4, 2>(buf, N, HW); run<STORE, 4, 3>(buf, N, HW); run<STORE, 4, 4>(buf, N, HW);
  run<STORE, 8, 0>(buf, N, HW); run<STORE, 8, 2>(buf, N, HW); run<STORE, 8, 3>(buf, N, HW); run<STORE, 8, 4>(buf, N, HW);
  run<STORE, 16, 0>(buf, N, HW); run<STORE, 16, 1>(buf, N, HW); run<STORE, 16, 2>(buf, N, HW); run<STORE, 16, 3>(buf, N, HW); run<STORE, 16, 4>(buf, N, HW);
}

int main() {
  const long N = 8, HW = 2048L * 2048L;
  float* buf;
  CK(hipMalloc(&buf, N * 16 * HW * 4));
  CK(hipMemset(buf, 0, N * 16 * HW * 4));
  all<true>(buf, N, HW);
  all<false>(buf, N, HW);
  return 0;
}
