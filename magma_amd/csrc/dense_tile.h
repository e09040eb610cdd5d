// 64 x 64 FP64 tile held in MFMA accumulators by a 256-thread workgroup (2 x 2 wavefronts, each
// 2 x 2 v_mfma_f64_16x16x4_f64 blocks): shared by dense.hip and dense_dag.hip.
#pragma once
#include "common.h"
#include "mfma_sweep.h"

namespace magma {

constexpr int TLD = 80;  // LDS row length (doubles) of a staged 16 x 64 operand slab
constexpr int NB = 64;

struct TileAcc {
  v4d c[2][2];  // [ni][mi]
};

__device__ __forceinline__ void tile_zero(TileAcc& acc) {
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc.c[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
}

// Visit every accumulator element of this thread: fn(m, n, value&).
template <class F>
__device__ __forceinline__ void tile_foreach(TileAcc& acc, F fn) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave & 1, wn = wave >> 1;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = wm * 32 + mi * 16 + (lane & 15);
        const int n = wn * 32 + ni * 16 + (lane >> 4) + 4 * r;
        fn(m, n, (double)acc.c[ni][mi][r]);
      }
}

}  // namespace magma
