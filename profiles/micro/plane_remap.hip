// Microbenchmark, follow-up of plane_order.hip: T = 1, planes innermost, layout unchanged; only the
// blockIdx.x -> chunk mapping changes, so that one XCD (which receives every 8th workgroup) no longer sees
// a single residue class of 4 KB chunks.
//   R0: chunk = b                                   (today)
//   R1: chunk = (b % 8) * (n / 8) + b / 8           XCD i owns a contiguous eighth of the plane
//   R2: chunk = b ^ ((b >> 3) & 7)                  xor-swizzle of the low three bits
//   R3: chunk = b ^ ((b >> 3) & 15)                 xor-swizzle of the low four bits
//   R4: chunk = (b % 8) * 64 + (b / 8) % 64 + (b / 512) * 512     XCD i owns 64-chunk (256 KB) strips
//   hipcc --offload-arch=gfx950 -O3 -o plane_remap plane_remap.hip && ./plane_remap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

template <bool STORE, int P, int R>
__global__ __launch_bounds__(256) void k(float* buf, long HW, int n, float* sink) {
  float* o = buf + long(blockIdx.y) * P * HW;
  const int b = blockIdx.x;
  int c = b;
  if (R == 1) c = (b % 8) * (n / 8) + b / 8;
  if (R == 2) c = b ^ ((b >> 3) & 7);
  if (R == 3) c = b ^ ((b >> 3) & 15);
  if (R == 4) c = (b % 8) * 64 + (b / 8) % 64 + (b / 512) * 512;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int p = 0; p < P; ++p) {
    float4* q = reinterpret_cast<float4*>(o + long(p) * HW + (long(c) * 256 + threadIdx.x) * 4);
    if (STORE) *q = make_float4(1.f, 2.f, 3.f, float(threadIdx.x));
    else { const float4 t = *q; acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w; }
  }
  if (!STORE && acc.x + acc.y + acc.z + acc.w == 12345.f) *sink = 1.f;
}

// interpolate-forward-like mix: read 4 planes (index + 3 bary) of `src`, write 16 planes of `buf`
template <int R>
__global__ __launch_bounds__(256) void kmix(float* buf, const float* src, long HW, int n) {
  float* o = buf + long(blockIdx.y) * 16 * HW;
  const float* q = src + long(blockIdx.y) * 4 * HW;
  const int b = blockIdx.x;
  int c = b;
  if (R == 4) c = (b % 8) * 64 + (b / 8) % 64 + (b / 512) * 512;
  const long off = (long(c) * 256 + threadIdx.x) * 4;
  float4 a = *reinterpret_cast<const float4*>(q + off);
#pragma unroll
  for (int p = 1; p < 4; ++p) {
    const float4 t = *reinterpret_cast<const float4*>(q + long(p) * HW + off);
    a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
  }
#pragma unroll
  for (int p = 0; p < 16; ++p) *reinterpret_cast<float4*>(o + long(p) * HW + off) = a;
}

template <int R>
void runmix(float* buf, const float* src, long N, long HW) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int n = int(HW / 1024);
  const dim3 grid((unsigned)n, (unsigned)N);
  hipLaunchKernelGGL((kmix<R>), grid, dim3(256), 0, 0, buf, src, HW, n);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((kmix<R>), grid, dim3(256), 0, 0, buf, src, HW, n);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 10;
  printf("mix   load 4 + store 16 planes R%d : %.3f ms  %.2f TB/s\n", R, ms, double(N) * 20 * HW * 4 / ms * 1e-9);
}

template <bool STORE, int P, int R>
void run(float* buf, long N, long HW) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int n = int(HW / 1024);
  const dim3 grid((unsigned)n, (unsigned)(N * 16 / P));
  hipLaunchKernelGGL((k<STORE, P, R>), grid, dim3(256), 0, 0, buf, HW, n, buf);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<STORE, P, R>), grid, dim3(256), 0, 0, buf, HW, n, buf);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 10;
  printf("%s P=%2d R%d : %.3f ms  %.2f TB/s\n", STORE ? "store" : "load ", P, R, ms, double(N) * 16 * HW * 4 / ms * 1e-9);
}

template <bool STORE>
void all(float* buf, long N, long HW) {
  run<STORE, 1, 0>(buf, N, HW); run<STORE, 1, 1>(buf, N, HW); run<STORE, 1, 2>(buf, N, HW);
  run<STORE, 4, 0>(buf, N, HW); run<STORE, 4, 1>(buf, N, HW); run<STORE, 4, 2>(buf, N, HW);
  run<STORE, 16, 0>(buf, N, HW); run<STORE, 16, 1>(buf, N, HW); run<STORE, 16, 2>(buf, N, HW); run<STORE, 16, 3>(buf, N, HW); run<STORE, 16, 4>(buf, N, HW);
}

int main() {
  const long N = 8, HW = 2048L * 2048L;
  float* buf;
  CK(hipMalloc(&buf, N * 16 * HW * 4));
  CK(hipMemset(buf, 0, N * 16 * HW * 4));
  all<true>(buf, N, HW);
  all<false>(buf, N, HW);
  float* src;
  CK(hipMalloc(&src, N * 4 * HW * 4));
  CK(hipMemset(src, 0, N * 4 * HW * 4));
  runmix<0>(buf, src, N, HW);
  runmix<4>(buf, src, N, HW);
  return 0;
}
