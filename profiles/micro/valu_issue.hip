// Microbenchmark: how many cycles does a SIMD of gfx950 spend per wave64 VALU instruction?
//
// Every "instruction-bound" floor in DESIGN.md / profiles/NOTES.md up to round 4 priced a wave64 VALU instruction at FOUR
// cycles (SIMD-16, the GCN lineage); /opt/skills/guides/MI355X_MICROARCH.md says TWO (SIMD-32).  This settles it, for the
// instruction kinds the path's kernels are made of, at 1 / 2 / 4 / 8 waves per SIMD:
//   fma32      v_fma_f32, 16 independent accumulators            mul+add    v_mul_f32 / v_add_f32 alternating
//   pk_mul     v_pk_mul_f32 (two floats per lane)                pk_add     v_pk_add_f32        pk_fma   v_pk_fma_f32
//   fma64      v_fma_f64          add64  v_add_f64      mul64  v_mul_f64      cvt64  v_cvt_f64_f32   cvt32  v_cvt_f32_f64
//   rcp        v_rcp_f32 (transcendental)       sqrt  v_sqrt_f32
//   iadd       v_add_u32          mad24  v_mad_u32_u24   lshladd  v_lshl_add_u32    mul_lo  v_mul_lo_u32
//   cmpsel     v_cmp_lt_f32 + v_cndmask_b32 pairs               dpp        v_mov_b32 row_newbcast (the rasterizer's row broadcast)
//   fma+salu   v_fma_f32 interleaved 1:1 with s_add_u32         dep        ONE dependent v_fma_f32 chain (latency)
//   min3       v_min3_f32         med3  v_med3_f32
// A wave runs ITERS x 128 instructions of its kind between two reads of s_memtime (shader cycles) and s_memrealtime (the
// constant 100 MHz counter): their ratio is the clock the chip actually ran at (it clocks down under dense VALU load), the
// table gives wave-instructions per SHADER cycle per SIMD -- the issue model -- and per nominal 2.4 GHz cycle from the
// kernel's wall time -- what a time estimate must be made with.  Placement: a workgroup asks for so much LDS that exactly
// one (1-4 waves per SIMD) or two (8) fit a CU and the grid is exactly what the chip holds, so every SIMD hosts the stated
// number of waves (a first version without that let the dispatcher double up CUs: wall and wave clocks disagreed by 1.5x).
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

enum Kind { FMA32, MULADD, PK_MUL, PK_ADD, PK_FMA, FMA64, ADD64, MUL64, CVT64, CVT32, RCP, SQRT, IADD, MAD24, LSHLADD, MULLO, CMPSEL, DPP, FMA_SALU, DEP, MIN3, MED3, NKIND };
static const char* kNames[NKIND] = {"fma32", "mul+add", "pk_mul", "pk_add", "pk_fma", "fma64", "add64", "mul64", "cvt64", "cvt32", "rcp", "sqrt", "iadd", "mad24", "lshladd", "mul_lo", "cmpsel", "dpp", "fma+salu", "dep", "min3", "med3"};
// VALU instructions in one block of 16 statements (cmpsel is two per statement, fma+salu counts the VALU half)
static const int kPerStmt[NKIND] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 1, 1, 1, 1};

#define R16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

template <int KIND>
__global__ __launch_bounds__(1024, 8) void k(unsigned long long* cycles, float* sink, int iters, float seed) {
  extern __shared__ float s_pad[]; // placement only
  if (seed == 12345.f) s_pad[threadIdx.x] = seed;
  float a[16];
  f2 p[16];
  double d[16];
  unsigned u[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = seed + float(i) + float(threadIdx.x) * 1e-3f;
    p[i] = f2{a[i], a[i] * 0.5f};
    d[i] = double(a[i]);
    u[i] = unsigned(i) + threadIdx.x;
  }
  float x = seed * 0.999f, y = 1.0f - seed * 1e-6f;
  f2 px = f2{x, y};
  double dx = double(x), dy = double(y);
  unsigned ux = 3u;
  unsigned sacc = 0;
  __syncthreads();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep) {
      if constexpr (KIND == FMA32) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(x));
        R16(M)
#undef M
      } else if constexpr (KIND == MULADD) {
#define M(i) if ((i) & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x)); else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
        R16(M)
#undef M
      } else if constexpr (KIND == PK_MUL) {
#define M(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(px));
        R16(M)
#undef M
      } else if constexpr (KIND == PK_ADD) {
#define M(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(px));
        R16(M)
#undef M
      } else if constexpr (KIND == PK_FMA) {
#define M(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(px));
        R16(M)
#undef M
      } else if constexpr (KIND == FMA64) {
#define M(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(dy), "v"(dx));
        R16(M)
#undef M
      } else if constexpr (KIND == ADD64) {
#define M(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dx));
        R16(M)
#undef M
      } else if constexpr (KIND == MUL64) {
#define M(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dy));
        R16(M)
#undef M
      } else if constexpr (KIND == CVT64) {
#define M(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(a[i]));
        R16(M)
#undef M
      } else if constexpr (KIND == CVT32) {
#define M(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(d[i]));
        R16(M)
#undef M
      } else if constexpr (KIND == RCP) {
#define M(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        R16(M)
#undef M
      } else if constexpr (KIND == SQRT) {
#define M(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
        R16(M)
#undef M
      } else if constexpr (KIND == IADD) {
#define M(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ux));
        R16(M)
#undef M
      } else if constexpr (KIND == MAD24) {
#define M(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(u[i]) : "v"(ux));
        R16(M)
#undef M
      } else if constexpr (KIND == LSHLADD) {
#define M(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[i]) : "v"(ux));
        R16(M)
#undef M
      } else if constexpr (KIND == MULLO) {
#define M(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(ux));
        R16(M)
#undef M
      } else if constexpr (KIND == CMPSEL) {
#define M(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
        R16(M)
#undef M
      } else if constexpr (KIND == DPP) {
#define M(i) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(a[((i) + 1) & 15]));
        R16(M)
#undef M
      } else if constexpr (KIND == FMA_SALU) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_add_u32 %1, %1, 1" : "+v"(a[i]), "+s"(sacc) : "v"(y), "v"(x) : "scc");
        R16(M)
#undef M
      } else if constexpr (KIND == DEP) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(y), "v"(x));
        R16(M)
#undef M
      } else if constexpr (KIND == MIN3) {
#define M(i) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(x));
        R16(M)
#undef M
      } else if constexpr (KIND == MED3) {
#define M(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(x));
        R16(M)
#undef M
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = float(sacc);
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + float(d[i]) + float(u[i]);
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 6;
    cycles[2 * w] = t1 - t0, cycles[2 * w + 1] = r1 - r0;
  }
}

template <int KIND>
void run(unsigned long long* d_cycles, float* sink, int cus) {
  const int iters = 2000;
  printf("%-9s", kNames[KIND]);
  for (int wps : {1, 2, 4, 8}) {
    // wps waves per SIMD: blocks of min(1024, 256 * wps) threads (a block's waves go round the CU's four SIMDs), one per
    // CU by LDS (100 KB each), or two of 1024 threads (70 KB each) for 8
    const int threads = std::min(1024, 256 * wps), blocks = cus * (wps == 8 ? 2 : 1);
    const size_t lds = wps == 8 ? 70 * 1024 : 100 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), lds, 0, d_cycles, sink, 10, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), lds, 0, d_cycles, sink, iters, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const int waves = blocks * threads / 64;
    std::vector<unsigned long long> c(2 * waves), cyc(waves), rt(waves);
    CK(hipMemcpy(c.data(), d_cycles, 2 * waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int i = 0; i < waves; ++i) cyc[i] = c[2 * i], rt[i] = c[2 * i + 1];
    std::sort(cyc.begin(), cyc.end());
    std::sort(rt.begin(), rt.end());
    const double med = double(cyc[waves / 2]);
    const double ghz = med / (double(rt[waves / 2]) * 10.0); // 100 MHz ticks -> ns
    const double per_wave = double(iters) * 128.0 * kPerStmt[KIND];
    // a SIMD hosts wps waves, each issuing per_wave instructions in `med` of its own cycles
    const double ipc_clock = per_wave * wps / med;
    const double ipc_wall = per_wave * waves / (double(cus) * 4.0) / (ms * 1e-3 * 2.4e9);
    printf("  | %dw: %.3f /cyc = %.2f cyc/inst, %.3f /2.4GHz-cyc, clock %.2f GHz", wps, ipc_clock, 1.0 / ipc_clock, ipc_wall, ghz);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
  }
  printf("\n");
}

template <int K0>
void run_all(unsigned long long* d_cycles, float* sink, int cus) {
  if constexpr (K0 < NKIND) {
    run<K0>(d_cycles, sink, cus);
    run_all<K0 + 1>(d_cycles, sink, cus);
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %s, %d CUs, clockRate %d kHz.  Wave-instructions per SIMD: per shader cycle (s_memtime), per nominal 2.4 GHz cycle (wall time); clock = s_memtime / s_memrealtime\n", prop.gcnArchName, cus, prop.clockRate);
  unsigned long long* d_cycles;
  float* sink;
  CK(hipMalloc(&d_cycles, size_t(cus) * 2 * 16 * 2 * sizeof(unsigned long long)));
  CK(hipMalloc(&sink, 64));
  run_all<0>(d_cycles, sink, cus);
  return 0;
}
