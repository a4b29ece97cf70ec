// Microbenchmark: how fast can a wave accumulate floats into an LDS table?  The scatter kernels (render backward,
// edge_scatter_pairs, interpolate backward C <= 4) end in "tail lanes add J values into a per-wave vertex table";
// this measures the candidates for that step, per CU (256-thread workgroups, every CU busy, 8 waves per SIMD):
//   A  ds_add_f32 (no return), L active lanes per instruction, distinct addresses
//   B  the same with all active lanes on ONE address (same-address conflicts)
//   C  ds_add_u32 (integer), distinct addresses
//   D  plain read - add - write (ds_read_b32, v_add_f32, ds_write_b32), distinct addresses (only valid when no two lanes
//      of the instruction share an address)
//   E  ds_add_rtn_f32 (returning)
//   F  ds_add_f64
//   G  ds_add_f64 with the 64 lanes in groups of g on one address each (64 / g distinct addresses per instruction): what
//      the sampler's texture-gradient windows see when neighbouring pixels hit the same texel
// Output: nanoseconds per wave-instruction and lane-operations per clock per CU at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomics lds_atomics.hip && ./lds_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int active, int same, int group = 1) {
  __shared__ float tab[4][1024];
  __shared__ double tabd[4][256];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = lane; i < 1024; i += 64) tab[wave][i] = 0.f;
  for (int i = lane; i < 256; i += 64) tabd[wave][i] = 0.0;
  __syncthreads();
  using LdsF = __attribute__((address_space(3))) float*;
  using LdsU = __attribute__((address_space(3))) unsigned*;
  using LdsD = __attribute__((address_space(3))) double*;
  float acc = 0.f;
  if (lane < active) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int a = same ? ((i + j) & 1023) : ((lane * 17 + (i * 8 + j) * 5) & 1023);  // 17: odd stride, conflict-free banks
        if (MODE == 0) __hip_atomic_fetch_add((LdsF)&tab[wave][a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) __hip_atomic_fetch_add((LdsU)&tab[wave][a], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) { LdsF q = (LdsF)&tab[wave][a]; *q = *q + 1.0f; }
        if (MODE == 4) acc += __hip_atomic_fetch_add((LdsF)&tab[wave][a], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 6) {
          const int ag = ((lane / group) * 17 + (i * 8 + j) * 5) & 255;
          __hip_atomic_fetch_add((LdsD)&tabd[wave][ag], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (MODE == 5) __hip_atomic_fetch_add((LdsD)&tabd[wave][a & 255], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  float s = acc;
  for (int i = lane; i < 1024; i += 64) s += tab[wave][i];
  for (int i = lane; i < 256; i += 64) s += float(tabd[wave][i]);
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
void run(const char* name, float* out, int active, int same, int group = 1) {
  const int iters = 2000, blocks = 256 * 8;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 10, active, same, group);
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, active, same, group);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double instr_per_cu = double(blocks) / 256 * 4 * iters * 8;  // wave-instructions issued on one CU
  const double ns = ms * 1e6 / instr_per_cu;
  printf("%-28s active %2d %s: %7.2f ns per wave-instruction per CU = %6.2f clocks -> %5.2f lane-ops / clock / CU\n", name, active,
         same ? "one address " : "distinct    ", ns, ns * 2.4, active / (ns * 2.4));
}

int main() {
  float* out;
  CK(hipMalloc(&out, 64));
  for (int active : {64, 32, 16, 8, 4, 1}) run<0>("A ds_add_f32", out, active, 0);
  for (int active : {64, 8, 2}) run<0>("B ds_add_f32", out, active, 1);
  for (int active : {64, 8}) run<2>("C ds_add_u32", out, active, 0);
  for (int active : {64, 8}) run<3>("D read-add-write", out, active, 0);
  for (int active : {64, 8}) run<4>("E ds_add_rtn_f32", out, active, 0);
  for (int active : {64, 8}) run<5>("F ds_add_f64", out, active, 0);
  for (int group : {1, 2, 4, 8, 16, 64}) {
    printf("group of %2d lanes per address: ", group);
    run<6>("G ds_add_f64", out, 64, 0, group);
  }
  return 0;
}
