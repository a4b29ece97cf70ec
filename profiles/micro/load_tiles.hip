// Microbenchmark: HBM read bandwidth of P = 16 channel planes of an [N,16,H,W] float tensor, by the SHAPE of the piece one
// wave instruction fetches -- the question behind interpolate backward's streaming floor (4.5-5.0 TB/s).
//   A  64 x 16 tile, 4 waves, wave = 4 rows of 64 px, 4 B per lane: 256-byte pieces, 16 planes x 4 rows per wave (the kernel's)
//   B  256 x 4 tile, 4 waves, wave = one row of 256 px, 16 B per lane: 1 KB contiguous per plane and wave instruction
//   C  256 x 16 tile, 16 waves (1024 threads), wave = one row of 256 px, 16 B per lane: 1 KB pieces, 16 rows per workgroup
//   E  64 x 16 tile as A, but 16 B per lane: lane l fetches pixels 4 (l % 16) .. + 3 of row l / 16 -- one instruction per plane
//      covers the wave's 4 rows x 64 px (four 256-byte pieces one image row apart)
//   F  64 x 16 tile, 16 B per lane: lane l fetches pixels 4 (l % 16) .. + 3 of PLANE 4 q + l / 16 of one row -- one instruction
//      covers four planes x 64 px (four 256-byte pieces one plane apart; measured inside the kernel in round 4: no gain)
//   G  128 x 8 tile, 4 waves, wave = 2 rows of 128 px, 16 B per lane (two 512-byte pieces per instruction)
// Each with the library's XCD strip order (tiles of 16 image rows per strip) and with the linear order.  Values are summed so nothing is optimised away.
//   hipcc --offload-arch=gfx950 -O3 -o load_tiles load_tiles.hip && ./load_tiles
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ int xcd_tile(int b, int n, int strip) {
  const int group = 8 * strip, base = b / group * group, r = b - base, left = n - base;
  if (left >= group) return base + (r % 8) * strip + r / 8;
  const int s = left / 8;
  if (r < s * 8) return base + (r % 8) * s + r / 8;
  return b;
}

constexpr int P = 16;

// MODE 0 = A, 1 = B, 2 = C, 3 = E, 4 = F, 5 = G
template <int MODE>
__global__ __launch_bounds__(MODE == 2 ? 1024 : 256) void k(const float* __restrict__ in, float* sink, int H, int W, int tiles_x, int strip) {
  const long HW = long(H) * W;
  const int n = blockIdx.y;
  const int tile = strip <= 1 ? int(blockIdx.x) : xcd_tile(blockIdx.x, gridDim.x, strip);
  const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* base = in + long(n) * P * HW;
  float acc = 0.f;
  if (MODE == 0) {
    const int x = tx * 64 + lane;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = ty * 16 + wave * 4 + r;
      const float* p0 = base + long(y) * W + x;
#pragma unroll
      for (int p = 0; p < P; ++p) acc += p0[long(p) * HW];
    }
  } else if (MODE == 3) {
    const int y = ty * 16 + wave * 4 + (lane >> 4);
    const int x = tx * 64 + (lane & 15) * 4;
    const float* p0 = base + long(y) * W + x;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float4 v = *reinterpret_cast<const float4*>(p0 + long(p) * HW);
      acc += v.x + v.y + v.z + v.w;
    }
  } else if (MODE == 4) {
    const int x = tx * 64 + (lane & 15) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int y = ty * 16 + wave * 4 + r;
      const float* p0 = base + long(y) * W + x + long(lane >> 4) * HW;
#pragma unroll
      for (int q = 0; q < P / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p0 + long(4 * q) * HW);
        acc += v.x + v.y + v.z + v.w;
      }
    }
  } else if (MODE == 5) {
    const int y = ty * 8 + wave * 2 + (lane >> 5);
    const int x = tx * 128 + (lane & 31) * 4;
    const float* p0 = base + long(y) * W + x;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float4 v = *reinterpret_cast<const float4*>(p0 + long(p) * HW);
      acc += v.x + v.y + v.z + v.w;
    }
  } else {
    const int rows = MODE == 1 ? 4 : 16;
    const int y = ty * rows + wave;
    const int x = tx * 256 + lane * 4;
    const float* p0 = base + long(y) * W + x;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float4 v = *reinterpret_cast<const float4*>(p0 + long(p) * HW);
      acc += v.x + v.y + v.z + v.w;
    }
  }
  if (acc == 12345.678f) sink[0] = acc;
}

template <int MODE>
void run(const float* buf, float* sink, int N, int H, int W, const char* name) {
  const int tw = (MODE == 0 || MODE == 3 || MODE == 4) ? 64 : (MODE == 5 ? 128 : 256), th = MODE == 1 ? 4 : (MODE == 5 ? 8 : 16);
  const int tiles_x = W / tw, tiles_y = H / th;
  const int strip = tiles_x * (16 / th);  // the tiles of 16 image rows
  const dim3 grid(tiles_x * tiles_y, N), block(MODE == 2 ? 1024 : 256);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int s : {strip, 1}) {
    hipLaunchKernelGGL((k<MODE>), grid, block, 0, 0, buf, sink, H, W, tiles_x, s);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<MODE>), grid, block, 0, 0, buf, sink, H, W, tiles_x, s);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 10;
    printf("%-58s %s  %.3f ms  %.2f TB/s\n", name, s == 1 ? "linear order" : "XCD strips  ", ms, double(N) * P * H * W * 4 / ms * 1e-9);
  }
}

int main() {
  const int N = 8, H = 2048, W = 2048;
  float *buf, *sink;
  CK(hipMalloc(&buf, size_t(N) * P * H * W * 4));
  CK(hipMalloc(&sink, 16));
  CK(hipMemset(buf, 0, size_t(N) * P * H * W * 4));
  run<0>(buf, sink, N, H, W, "A  64 x 16 tile, 4 waves x 4 rows,  4 B/lane (256 B pieces)");
  run<1>(buf, sink, N, H, W, "B 256 x  4 tile, 4 waves x 1 row,  16 B/lane (1 KB pieces)");
  run<2>(buf, sink, N, H, W, "C 256 x 16 tile, 16 waves x 1 row, 16 B/lane (1 KB pieces)");
  run<3>(buf, sink, N, H, W, "E  64 x 16 tile, 4 waves x 4 rows, 16 B/lane (4 rows x 256 B)");
  run<4>(buf, sink, N, H, W, "F  64 x 16 tile, 4 waves x 4 rows, 16 B/lane (4 planes x 256 B)");
  run<5>(buf, sink, N, H, W, "G 128 x  8 tile, 4 waves x 2 rows, 16 B/lane (2 rows x 512 B)");
  return 0;
}
