// Microbenchmark: float atomic-add request rate, agent scope into ONE buffer vs workgroup scope into
// a per-XCD private copy (copy chosen by HW_REG_XCC_ID at run time, so only CUs that share an L2
// ever touch a copy).  A request = 16 consecutive floats (64 B), 4 requests per wave instruction.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_scope atomic_scope.hip && ./atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 15u;
}
__device__ inline unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

template <int MODE>  // 0 agent one buffer, 1 workgroup-scope per-XCD copy, 2 agent-scope per-XCD copy
__global__ __launch_bounds__(256) void k(float* buf, int V, int iters, size_t copy_stride) {
  const unsigned wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const unsigned lane = threadIdx.x & 63;
  float* base = buf;
  if (MODE != 0) base += size_t(xcc_id()) * copy_stride;
  for (int i = 0; i < iters; ++i) {
    const unsigned v = hash32(wave * 7919u + i * 104729u + (lane >> 4) * 31u) % unsigned(V);
    float* p = base + size_t(v) * 16 + (lane & 15);
    if (MODE == 1)
      __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else
      __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void reduce8(const float* copies, float* out, size_t n, size_t stride) {
  size_t i = size_t(blockIdx.x) * 256 + threadIdx.x;
  if (i >= n) return;
  float s = 0;
  for (int c = 0; c < 8; ++c) s += copies[c * stride + i];
  out[i] = s;
}

int main() {
  const int V = 50400 * 8, iters = 64, blocks = 256 * 16;
  const size_t n = size_t(V) * 16;
  float *one, *copies, *red;
  CK(hipMalloc(&one, n * 4));
  CK(hipMalloc(&copies, n * 4 * 8));
  CK(hipMalloc(&red, n * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double requests = double(blocks) * 4 * iters * 4;
  std::vector<float> a(n), b(n);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(one, 0, n * 4));
      CK(hipMemset(copies, 0, n * 4 * 8));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, one, V, iters, n);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, copies, V, iters, n);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, copies, V, iters, n);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("mode %d rep %d: %.3f ms  %.2f G requests/s\n", mode, rep, ms, requests / ms * 1e-6);
    }
    if (mode == 0) CK(hipMemcpy(a.data(), one, n * 4, hipMemcpyDeviceToHost));
    if (mode >= 1) {
      hipLaunchKernelGGL(reduce8, dim3((n + 255) / 256), dim3(256), 0, 0, copies, red, n, n);
      CK(hipMemcpy(b.data(), red, n * 4, hipMemcpyDeviceToHost));
      size_t bad = 0;
      double tot = 0;
      for (size_t i = 0; i < n; ++i) { bad += a[i] != b[i]; tot += b[i]; }
      printf("mode %d: sum-of-copies vs agent single buffer: %zu mismatches, total %.0f (expect %.0f)\n", mode, bad, tot, requests * 16);
    }
  }
  return 0;
}
