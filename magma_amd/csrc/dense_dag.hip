// Dense SPD inverse + log-det as ONE launch: the tile tasks of a left-looking blocked Cholesky,
// of the triangular inverse and of A^-1 = L^-T L^-1 form a dependency graph that persistent
// workgroups walk in ticket order, handing tiles to each other through memory inside the launch.
//
// Why: at N = 2000 the launch-per-step factorisation (dense.hip) is a chain of 2 x 32 dependent
// launches + 11 for the triangular inverse, each paying launch, fill and drain; the matrix cores
// idle (4-5 % of FP64 peak). Here a step of the chain is: diagonal tile factored -> flag ->
// tile below it finished -> flag -> next diagonal tile, while every other tile has already
// accumulated all but its last product, the rows of L^-T follow one step behind the
// factorisation, and A^-1's tiles accumulate as the rows they need appear.
//
// Tasks (nt = Np / 64 block rows), in ticket order -- a task depends only on earlier tickets, so a
// workgroup that spins on a flag always waits for a workgroup that is already running, whatever
// the grid size and whatever else occupies the chip (no co-residency assumption):
//   for j = 0 .. nt-1:
//     D(j)          L_{j,j-1} (as F below), then S = A_jj - sum_{k<j} L_jk L_jk^T;
//                   X_jj = chol(S)^-1, U_jj = X_jj^T
//     F(i,j), i>j+1 L_ij = (A_ij - sum_{k<j} L_ik L_jk^T) X_jj^T
//     T(c,j), c<j   U_cj = -(sum_{c<=k<j} U_ck L_jk^T) X_jj^T        (U = L^-T, upper triangular)
//   I(i,j), i>=j    Ainv_ij = sum_{k>=i} U_ik U_jk^T, mirrored
// Hand-off (guide: MI355X_MICROARCH, inter-workgroup visibility, first row of the table): tiles
// another workgroup reads are written with write-through 16-byte stores (sc1), every storing
// wavefront drains its stores, a workgroup barrier, then ONE lane raises the tile's flag with an
// sc1 store; consumers poll flags with sc1 loads from one wavefront, pass a workgroup barrier,
// and load the tile with sc1 loads only. L, U and the flags live in the caller's workspace and
// are never read before they are written in a launch; the input A is only read, the output is
// written by the I tasks, which finish after every read of A (plain stores: the next kernel on
// the stream reads them).
// Every spin is bounded: a workgroup that waits ~seconds raises the abort word, every workgroup
// leaves at its next poll or ticket, and info is set to -999.
#include <type_traits>
#include "common.h"
#include "kernels.h"
#include "mfma_sweep.h"
#include "dense_tile.h"

namespace magma {

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

#define MAGMA_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
constexpr int AUX_SC1 = 16;
constexpr unsigned SPIN_LIMIT = 1u << 22;

struct DagArgs {
  const double* A;  // input (lower tiles read) and output (full symmetric inverse)
  double* Aout;
  double* L;   // strictly lower tiles: L_ij; diagonal tiles: X_jj = L_jj^-1
  double* U;   // upper tiles incl. diagonal: U = L^-T
  int* fL;     // nt x nt flags, [i * nt + k]: tile (i, k) of L published (k == i: X_ii and U_ii)
  int* fU;     // nt x nt flags, [c * nt + k]: tile (c, k) of U published (k == c: same as fL)
  int* ctrl;   // [0] ticket counter, [1] abort word, [2], [3] diagnostics, [4] walker claim
  int* fH;     // nt flags (walker mode): S~_jj and B_j of block step j parked in L(j,j) / L(j,j-1)
  int* fS;     // nt flags (walker 2): S~_jj parked by its own task HS(j); fH[j] then announces B_j only
  int* fV;     // nt flags (walker 2): V_{j,j-2} = A_{j,j-2} - sum_{k<j-2} L_jk L_{j-2,k}^T parked in tile (j, j-2) of U
  int walker;  // 1, 2: one workgroup (the first to start) walks the chain of diagonal tiles (dag_walker / dag_walker2)
  double* logdet;
  int* info;
  long long* stamps;  // diagnostic (MAGMA_DAG_STAMPS builds): 8 wall-clock stamps per ticket
  int nt;
  int ld;  // = Np (elements); every tile offset fits 32 bits of BYTES (checked by the launcher)
};

__device__ __forceinline__ rsrc_t make_rsrc(const double* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<double*>(p), 0, (int)bytes, 0x00020000);
}

// Wait until the flags fA[k..] and fB[k..] are up for at least tile k; returns the first tile
// index (<= k1) not known to be published, or -1 when the launch is being abandoned. Called by
// every thread of the workgroup (contains barriers); the first wavefront polls up to 32 flags of
// each row at once so that a tile whose operands were published long ago polls once.
__device__ __forceinline__ int poll_ready(const int* fA, const int* fB, int k, int k1, int* ctrl,
                                          int* s_ready) {
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x;
    const int kk = k + (lane & 31);
    const int* f = (lane < 32) ? fA : fB;
    int r = 0;
    for (unsigned spins = 0;; ++spins) {
      const int v = (kk < k1) ? __hip_atomic_load(f + kk, MAGMA_RLX_AGENT) : 1;
      const unsigned long long m = __ballot(v != 0);
      const unsigned lo = (unsigned)m, hi = (unsigned)(m >> 32);
      const int ra = (lo == 0xffffffffu) ? 32 : __builtin_ctz(~lo);
      const int rb = (hi == 0xffffffffu) ? 32 : __builtin_ctz(~hi);
      r = ra < rb ? ra : rb;
      if (r > 0) break;
      __builtin_amdgcn_s_sleep(8);
      if ((spins & 255u) == 255u) {
        int ab = __hip_atomic_load(ctrl + 1, MAGMA_RLX_AGENT);
        if (spins >= SPIN_LIMIT) {
          ab = 1;
          if (lane == 0) {
            // diagnostic: which flag this workgroup gave up on (index relative to ctrl), who
            __hip_atomic_store(ctrl + 3, (int)((fA + k) - ctrl) | ((int)blockIdx.x << 16), MAGMA_RLX_AGENT);
            __hip_atomic_store(ctrl + 1, 1, MAGMA_RLX_AGENT);
          }
        }
        if (ab != 0) {
          r = -1;
          break;
        }
      }
    }
    if (lane == 0) {
      int rdy = k + r;
      if (rdy > k1) rdy = k1;
      *s_ready = (r > 0) ? rdy : -1;
    }
  }
  __syncthreads();
  const int ready = *s_ready;
  __syncthreads();
  return ready;
}

// acc(m, n) += sum over tiles k0 <= k < k1 of sum_kk Arow(m, 64 k + kk) Brow(n, 64 k + kk),
// Arow / Brow = block rows (64 x Np, column-major) at ELEMENT offsets offA / offB of the buffers
// behind ra / rb; tile k of each row is loaded only after its flag (fA[k], fB[k]) is up.
// A whole 64 x 64 tile of each operand is fetched at once (16 loads of 16 bytes per thread in
// flight: one memory latency per tile, which is what the chain of diagonal tiles sees) and fed
// to LDS in four 16-column slabs; the next tile's loads are issued under the last slab's MFMAs.
// Returns false when the launch is being abandoned.
// With SELF, accS(m, n) += sum Arow(m, .) Arow(n, .) as well, from the same staged slabs (the
// diagonal tile of block row A next to its tile in block column B: D(j) below).
template <bool SELF>
__device__ __forceinline__ bool acc_rows_t(rsrc_t ra, unsigned offA, const int* fA, rsrc_t rb,
                                           unsigned offB, const int* fB, int k0, int k1, int ld,
                                           TileAcc& acc, TileAcc& accS, double* As, double* Bs,
                                           double* S2, int* ctrl, int* s_ready) {
  if (k0 >= k1) return true;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int x4 = (tid & 15) * 4, kr = tid >> 4;
  const int fo = (lane >> 4) * TLD + (lane & 15);
  int ready = poll_ready(fA, fB, k0, k1, ctrl, s_ready);
  if (ready < 0) return false;
  v4u ta[4][2], tb[4][2];
  auto fetch_slab = [&](int k, int s) {
    const unsigned col = (unsigned)(64 * k + 16 * s + kr) * (unsigned)ld;
    const unsigned oa = (offA + x4 + col) * 8u, ob = (offB + x4 + col) * 8u;
    ta[s][0] = __builtin_amdgcn_raw_buffer_load_b128(ra, oa, 0, AUX_SC1);
    ta[s][1] = __builtin_amdgcn_raw_buffer_load_b128(ra, oa + 16u, 0, AUX_SC1);
    tb[s][0] = __builtin_amdgcn_raw_buffer_load_b128(rb, ob, 0, AUX_SC1);
    tb[s][1] = __builtin_amdgcn_raw_buffer_load_b128(rb, ob + 16u, 0, AUX_SC1);
  };
  auto fetch = [&](int k) {
#pragma unroll
    for (int s = 0; s < 4; ++s) fetch_slab(k, s);
  };
  fetch(k0);
  // Two slab buffers (the second pair in S2, LDS the tile tasks do not otherwise use). The slabs
  // form one pipeline that runs on across tiles while the next tile is known to be published, and
  // everything that is not an MFMA is placed BETWEEN the 16 MFMAs of a slab (one wavefront per
  // SIMD issues in order: what follows a run of MFMAs in program order waits for all of them to
  // issue, 64 cycles each, and measured that way a tile product was 1.9 us of staging, barriers
  // and fragment reads PLUS 1.4 us of MFMAs): slab g + 1 goes to LDS under the first four MFMAs
  // of slab g, fragment groups 2 and 3 of slab g are read under the next two, the next tile's
  // loads go out under the two after, the barrier that makes slab g + 1 visible sits in front of
  // the last group, and groups 0 and 1 of slab g + 1 are read under that group. Same MFMAs on
  // the same operands in the same order: results are bit-identical.
  double* Ab[2] = {As, S2};
  double* Bb[2] = {Bs, S2 + 16 * TLD};
  double fa[2][4][2], fb[2][4][2], fs[2][4][2];
  auto stage_part = [&](int s, int part) {  // registers of slab s -> buffer s & 1, in four parts
    double* An = Ab[s & 1];
    double* Bn = Bb[s & 1];
    if (part == 0) *reinterpret_cast<v4u*>(&An[kr * TLD + x4]) = ta[s][0];
    if (part == 1) *reinterpret_cast<v4u*>(&An[kr * TLD + x4 + 2]) = ta[s][1];
    if (part == 2) *reinterpret_cast<v4u*>(&Bn[kr * TLD + x4]) = tb[s][0];
    if (part == 3) *reinterpret_cast<v4u*>(&Bn[kr * TLD + x4 + 2]) = tb[s][1];
  };
  auto read_group = [&](int p, int q) {  // k-group q of the slab in buffer p -> fragment set p
    const double* Ac = Ab[p];
    const double* Bc = Bb[p];
    const int kk = 4 * q;
    fa[p][q][0] = Ac[kk * TLD + fo + wm * 32];
    fa[p][q][1] = Ac[kk * TLD + fo + wm * 32 + 16];
    fb[p][q][0] = Bc[kk * TLD + fo + wn * 32];
    fb[p][q][1] = Bc[kk * TLD + fo + wn * 32 + 16];
    if constexpr (SELF) {
      fs[p][q][0] = Ac[kk * TLD + fo + wn * 32];
      fs[p][q][1] = Ac[kk * TLD + fo + wn * 32 + 16];
    }
  };
  // MFMA e (0..3) of k-group q; with SELF the two accumulators' MFMAs alternate
  auto mfma1 = [&](int p, int q, int e) {
    const int ni = e >> 1, mi = e & 1;
    acc.c[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[p][q][ni], fa[p][q][mi], acc.c[ni][mi], 0, 0, 0);
  };
  auto mfmaS = [&](int p, int q, int e) {
    if constexpr (SELF) {
      const int ni = e >> 1, mi = e & 1;
      accS.c[ni][mi] = __builtin_amdgcn_mfma_f64_16x16x4f64(fs[p][q][ni], fa[p][q][mi], accS.c[ni][mi], 0, 0, 0);
    }
  };
#define MAGMA_SB() __builtin_amdgcn_sched_barrier(0)
  // One tile. PF: tile k + 1 is known to be published -- its loads go out as this tile's registers
  // come free and its first slab is staged under this tile's last one, so that the pipeline runs
  // on (a separate, branch-free instantiation: with the loads behind run-time conditions the
  // compiler's wait-count pass falls back to vmcnt(0) in front of the staging stores, i.e. waits
  // for the loads it issued one slab ago).
  auto tile = [&](auto pf_tag, int k) {
    constexpr bool PF = decltype(pf_tag)::value;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int p = s & 1;
      const bool follows = (s < 3) || PF;  // another slab of the pipeline follows this one
      const int sn = (s + 1) & 3;          // its index in ta / tb (slab 0 of tile k + 1 after slab 3)
      MAGMA_SB();
      // group 0: the next slab goes to LDS (its buffer was last read before the previous barrier)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mfma1(p, 0, e);
        if (follows) stage_part(sn, e);
        MAGMA_SB();
      }
      if constexpr (SELF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mfmaS(p, 0, e);
        MAGMA_SB();
      }
      // group 1: this slab's groups 2 and 3 come out of LDS; the next tile's loads go out as this
      // tile's registers come free
      mfma1(p, 1, 0);
      read_group(p, 2);
      MAGMA_SB();
      mfma1(p, 1, 1);
      read_group(p, 3);
      MAGMA_SB();
      mfma1(p, 1, 2);
      if (PF) {
        if (s == 0) fetch_slab(k + 1, 0);
        if (s == 1) fetch_slab(k + 1, 2);
        if (s == 2) fetch_slab(k + 1, 3);
      }
      MAGMA_SB();
      mfma1(p, 1, 3);
      if (PF && s == 0) fetch_slab(k + 1, 1);
      MAGMA_SB();
      if constexpr (SELF) {
#pragma unroll
        for (int e = 0; e < 4; ++e) mfmaS(p, 1, e);
        MAGMA_SB();
      }
      // group 2
#pragma unroll
      for (int e = 0; e < 4; ++e) mfma1(p, 2, e);
#pragma unroll
      for (int e = 0; e < 4; ++e) mfmaS(p, 2, e);
      MAGMA_SB();
      // the next slab is visible, and every wavefront has read this one
      if (follows) __syncthreads();
      MAGMA_SB();
      // group 3, with groups 0 and 1 of the next slab read under it
      mfma1(p, 3, 0);
      if (follows) read_group(p ^ 1, 0);
      MAGMA_SB();
      mfma1(p, 3, 1);
      if (follows) read_group(p ^ 1, 1);
      MAGMA_SB();
      mfma1(p, 3, 2);
      mfma1(p, 3, 3);
#pragma unroll
      for (int e = 0; e < 4; ++e) mfmaS(p, 3, e);
      MAGMA_SB();
    }
  };
  int k = k0;
#pragma unroll 1
  for (;;) {
    // (slab 0 of tile k: loads issued) -> LDS, visible; its groups 0 and 1 in fragment set 0
    __syncthreads();  // both buffers free (whatever read them last has passed this barrier)
#pragma unroll
    for (int part = 0; part < 4; ++part) stage_part(0, part);
    __syncthreads();
    read_group(0, 0);
    read_group(0, 1);
    const int kf = (ready < k1 ? ready : k1) - 1;  // tiles k .. kf - 1 have a published successor
#pragma unroll 1
    for (; k < kf; ++k) tile(std::true_type{}, k);
    tile(std::false_type{}, k);
    ++k;
    if (k >= k1) break;
    ready = poll_ready(fA, fB, k, k1, ctrl, s_ready);
    if (ready < 0) return false;
    fetch(k);
  }
#undef MAGMA_SB
  return true;
}

__device__ __forceinline__ bool acc_rows(rsrc_t ra, unsigned offA, const int* fA, rsrc_t rb,
                                         unsigned offB, const int* fB, int k0, int k1, int ld,
                                         TileAcc& acc, double* As, double* Bs, double* S2, int* ctrl,
                                         int* s_ready) {
  return acc_rows_t<false>(ra, offA, fA, rb, offB, fB, k0, k1, ld, acc, acc, As, Bs, S2, ctrl, s_ready);
}

// out(m, n) = sum_k V(m, k) X(n, k): V staged in LDS as Vs[k * TLD + m] (64 rows of TLD), X = the
// 64 x 64 tile at element offset offX of rx (column-major, ld), published behind flag fX.
__device__ __forceinline__ bool times_xt(const double* Vs, rsrc_t rx, unsigned offX, const int* fX,
                                         int ld, TileAcc& out, double* Bs, double* B2, int* ctrl,
                                         int* s_ready, long long* stamp2) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int x4 = (tid & 15) * 4, kr = tid >> 4;
  const int fo = (lane >> 4) * TLD + (lane & 15);
  if (poll_ready(fX, fX, 0, 1, ctrl, s_ready) < 0) return false;  // (also publishes Vs)
#ifdef MAGMA_DAG_STAMPS
  if (stamp2 != nullptr && threadIdx.x == 0) *stamp2 = (long long)wall_clock64();
#endif
  v4u tb[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const unsigned ob = (offX + x4 + (unsigned)(16 * s + kr) * (unsigned)ld) * 8u;
    tb[s][0] = __builtin_amdgcn_raw_buffer_load_b128(rx, ob, 0, AUX_SC1);
    tb[s][1] = __builtin_amdgcn_raw_buffer_load_b128(rx, ob + 16u, 0, AUX_SC1);
  }
  double* Bb[2] = {Bs, B2};  // slab s + 1 is written while slab s is multiplied (see acc_rows_t)
  __syncthreads();
  *reinterpret_cast<v4u*>(&Bb[0][kr * TLD + x4]) = tb[0][0];
  *reinterpret_cast<v4u*>(&Bb[0][kr * TLD + x4 + 2]) = tb[0][1];
  __syncthreads();
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (s + 1 < 4) {
      *reinterpret_cast<v4u*>(&Bb[(s + 1) & 1][kr * TLD + x4]) = tb[s + 1][0];
      *reinterpret_cast<v4u*>(&Bb[(s + 1) & 1][kr * TLD + x4 + 2]) = tb[s + 1][1];
    }
    const double* Va = Vs + 16 * s * TLD;
    const double* Bc = Bb[s & 1];
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
      const double fa0 = Va[kk * TLD + fo + wm * 32];
      const double fa1 = Va[kk * TLD + fo + wm * 32 + 16];
      const double fb0 = Bc[kk * TLD + fo + wn * 32];
      const double fb1 = Bc[kk * TLD + fo + wn * 32 + 16];
      out.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb0, fa0, out.c[0][0], 0, 0, 0);
      out.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb0, fa1, out.c[0][1], 0, 0, 0);
      out.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb1, fa0, out.c[1][0], 0, 0, 0);
      out.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb1, fa1, out.c[1][1], 0, 0, 0);
    }
    if (s + 1 < 4) __syncthreads();
  }
  return true;
}

// acc(m, n) += sum_{k < 64} Am[k * TLD + m] Bm[k * TLD + n], both operands staged in LDS in
// operand layout. The fragment reads run a batch of four k-groups ahead of the MFMAs: left alone
// the compiler puts each k-group's reads right in front of its MFMAs and waits for them, 16
// exposed LDS latencies per product on the chain of diagonal tiles.
__device__ __forceinline__ void lds_product64(const double* Am, const double* Bm, TileAcc& acc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int fo = (lane >> 4) * TLD + (lane & 15);
  double fa[2][4][2], fb[2][4][2];  // two batches of 4 k-groups in flight
  auto load = [&](int bt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kk = 16 * bt + 4 * q;
      fa[bt & 1][q][0] = Am[kk * TLD + fo + wm * 32];
      fa[bt & 1][q][1] = Am[kk * TLD + fo + wm * 32 + 16];
      fb[bt & 1][q][0] = Bm[kk * TLD + fo + wn * 32];
      fb[bt & 1][q][1] = Bm[kk * TLD + fo + wn * 32 + 16];
    }
  };
  load(0);
#pragma unroll
  for (int bt = 0; bt < 4; ++bt) {
    if (bt + 1 < 4) load(bt + 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double a0 = fa[bt & 1][q][0], a1 = fa[bt & 1][q][1];
      const double b0 = fb[bt & 1][q][0], b1 = fb[bt & 1][q][1];
      acc.c[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a0, acc.c[0][0], 0, 0, 0);
      acc.c[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, a1, acc.c[0][1], 0, 0, 0);
      acc.c[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a0, acc.c[1][0], 0, 0, 0);
      acc.c[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, a1, acc.c[1][1], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// acc <- -A(m, n) for the 64 x 64 tile at A (plain loads: the input is only read in this launch).
__device__ __forceinline__ void tile_load_neg(TileAcc& acc, const double* __restrict__ A, int ld) {
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
        acc.c[ni][mi][r] = -A[m + (int64_t)n * ld];
      }
}

// Publish the 64 x 64 tile staged in LDS as Ts[n * TLD + m] (column n, row m) to element offset
// off of rd: write-through 16-byte stores, 512 contiguous bytes per half wavefront.
__device__ __forceinline__ void store_tile_sc1(const double* Ts, rsrc_t rd, unsigned off, int ld) {
  const int tid = threadIdx.x;
  const int m2 = (tid & 31) * 2;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int n = c * 8 + (tid >> 5);
    const v4u v = *reinterpret_cast<const v4u*>(&Ts[n * TLD + m2]);
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, (off + m2 + (unsigned)n * (unsigned)ld) * 8u, 0,
                                           AUX_SC1);
  }
}

// Every storing wavefront has drained, then one lane raises the flag(s).
__device__ __forceinline__ void publish(int* f0, int* f1) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_store(f0, 1, MAGMA_RLX_AGENT);
    if (f1 != nullptr) __hip_atomic_store(f1, 1, MAGMA_RLX_AGENT);
  }
}

__device__ __forceinline__ void abandon(int* info) {
  if (threadIdx.x == 0) __hip_atomic_store(info, -999, MAGMA_RLX_AGENT);
}


#ifdef MAGMA_DAG_STAMPS
#define DAG_STAMP(t, s)                                                          \
  do {                                                                           \
    if (a.stamps != nullptr && threadIdx.x == 0) a.stamps[(t) * 8 + (s)] = (long long)wall_clock64(); \
  } while (0)
#define DAG_STAMP_T(t, s, who)                                                   \
  do {                                                                           \
    if (a.stamps != nullptr && threadIdx.x == (who)) a.stamps[(t) * 8 + (s)] = (long long)wall_clock64(); \
  } while (0)
#else
#define DAG_STAMP(t, s) do { } while (0)
#define DAG_STAMP_T(t, s, who) do { } while (0)
#endif

// The two tiles block step j of the walker needs (see dag_walker), requested by NTHR threads
// (q = index of this thread among them): 16-byte unit u = q + i NTHR of a tile = rows m2, m2 + 1
// of column n, m2 = 2 (u & 31), n = u >> 5.
template <int NTHR>
__device__ __forceinline__ void walker_fetch(int q, int j, rsrc_t rL, int ld,
                                             v4u (&tS)[2048 / NTHR], v4u (&tB)[2048 / NTHR]) {
  const unsigned offD = (unsigned)j * NB * (unsigned)(ld + 1);
  const unsigned offB = (unsigned)j * NB + (unsigned)(j > 0 ? j - 1 : 0) * NB * (unsigned)ld;
#pragma unroll
  for (int i = 0; i < 2048 / NTHR; ++i) {
    const int u = q + i * NTHR;
    const unsigned e = (unsigned)((u & 31) * 2) + (unsigned)(u >> 5) * (unsigned)ld;
    tS[i] = __builtin_amdgcn_raw_buffer_load_b128(rL, (offD + e) * 8u, 0, AUX_SC1);
    if (j > 0) tB[i] = __builtin_amdgcn_raw_buffer_load_b128(rL, (offB + e) * 8u, 0, AUX_SC1);
  }
}

// ... and parked in LDS: B_j as Ws[k * TLD + m] (operand layout), S~_jj as Ss[m * 65 + n].
template <int NTHR>
__device__ __forceinline__ void walker_park(int q, int j, const v4u (&tS)[2048 / NTHR],
                                            const v4u (&tB)[2048 / NTHR], double* Ws, double* Ss) {
#pragma unroll
  for (int i = 0; i < 2048 / NTHR; ++i) {
    const int u = q + i * NTHR;
    const int m2 = (u & 31) * 2, n = u >> 5;
    double2 pr;
    __builtin_memcpy(&pr, &tS[i], 16);
    Ss[m2 * 65 + n] = pr.x;
    Ss[(m2 + 1) * 65 + n] = pr.y;
    if (j > 0) *reinterpret_cast<v4u*>(&Ws[n * TLD + m2]) = tB[i];
  }
}

// The chain of diagonal tiles walked by ONE workgroup (outside the ticket order).
// With a task per block step (D(j) below) every step pays: X_{j-1,j-1} stored, drained, flagged,
// polled by the next step's workgroup and fetched back from L2 -- 3-4 of the 21 us a step takes.
// Here X stays in LDS from one step to the next. What a step needs from the rest of the graph,
//   S~_jj = A_jj - sum_{k<j-1} L_jk L_jk^T   and   B_j = A_{j,j-1} - sum_{k<j-1} L_jk L_{j-1,k}^T,
// is accumulated by the ticket task H(j) (the slot D(j) had), parked in the tiles L(j,j) /
// L(j,j-1) (nobody reads those before their flags go up) and announced through fH[j]. H(j + 1)
// gets its last term (L_{j,j-1}) early in step j and finishes under the factorisation of step j,
// during which the two wavefronts that have no part in it fetch its tiles into LDS. Step j:
//   L_{j,j-1} = B_j X_{j-1,j-1}^T (both in LDS) -> S = S~ - L L^T -> 64 x 64 factor -> X_jj, U_jj.
// The walker holds no ticket and waits only for H tasks; a worker only ever waits for earlier
// tickets or for the walker, which is running by the time any worker starts (see the kernel), so
// with one worker resident the graph drains.
__device__ __forceinline__ void dag_walker(const DagArgs& a, rsrc_t rL, rsrc_t rU, double* Vs,
                                           double* Ws, double* Ss, double* Pn, double* Xch,
                                           int* s_ready, int* s_walk) {
  const int nt = a.nt, ld = a.ld;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int l = lane & 15, g = lane >> 4;
  const int fo = (lane >> 4) * TLD + (lane & 15);
  const int m2 = (tid & 31) * 2;
  {  // step 0's tile, by everybody
    if (poll_ready(a.fH, a.fH, 0, 1, a.ctrl, s_ready) < 0) return abandon(a.info);
    v4u tS[8], tB[8];
    walker_fetch<256>(tid, 0, rL, ld, tS, tB);
    walker_park<256>(tid, 0, tS, tB, Ws, Ss);
    __syncthreads();
  }
  for (int j = 0; j < nt; ++j) {
    const unsigned offD = (unsigned)j * NB * (unsigned)(ld + 1);
    const unsigned offB = (unsigned)j * NB + (unsigned)(j > 0 ? j - 1 : 0) * NB * (unsigned)ld;
    const int ts = nt * nt + nt * (nt + 1) / 2 + j;  // (diagnostic stamps: slots behind the tickets')
    (void)ts;
    DAG_STAMP(ts, 0);
    TileAcc acc;
    tile_zero(acc);
    if (j > 0) {
      // B_j is in Ws[k * TLD + m], X_{j-1,j-1} in Vs[k * TLD + n] = X(n, k) since the last step
      TileAcc out;
      tile_zero(out);
      lds_product64(Ws, Vs, out);
      __syncthreads();  // every wavefront has read Ws and Vs
      DAG_STAMP(ts, 1);
      tile_foreach(out, [&](int m, int n, double v) { Ws[n * TLD + m] = v; });  // L_{j,j-1}
      __syncthreads();
      store_tile_sc1(Ws, rL, offB, ld);
      lds_product64(Ws, Ws, acc);  // acc = L L^T
    }
    // S = S~ - L L^T staged for the factor as Vs[m * 65 + n] (X of the last step is dead: every
    // wavefront passed the barrier after the product)
    tile_foreach(acc, [&](int m, int n, double v) { Vs[m * 65 + n] = Ss[m * 65 + n] - v; });
    // L_{j,j-1}'s flag: its stores have drained under the product above
    if (j > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (j > 0 && tid == 0) __hip_atomic_store(a.fL + j * nt + (j - 1), 1, MAGMA_RLX_AGENT);
    DAG_STAMP(ts, 2);
    if (wave == 0) {
      v4d S[4][4];
#pragma unroll
      for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int rr = 16 * it + g + 4 * v, cc = 16 * jt + l;
            S[it][jt][v] = (rr >= cc) ? Vs[rr * 65 + cc] : Vs[cc * 65 + rr];
          }
      chol64_factor(S, Pn, Xch, lane);
      __syncthreads();  // S was read before the factor's barriers; X may now overwrite it
    } else if (wave == 1) {
      v4d B[4][4];
      double pm = 1.0;
      int pe = 0, bad = 0;
      chol64_solve(B, Pn + 256, Xch, lane, pm, pe, bad);
      if (lane == 0) {
        atomicAdd(a.logdet, log(pm) + (double)pe * 0.69314718055994530941723212145818);
        if (bad != 0) atomicCAS(a.info, 0, j * NB + bad);
      }
      __syncthreads();
#pragma unroll
      for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int rr = 16 * it + g + 4 * v, cc = 16 * jt + l;
            const double x = (jt <= it) ? B[it][jt][v] : 0.0;  // X(rr, cc)
            Vs[cc * TLD + rr] = x;                              // column cc of X
            // U_jj = X_jj^T straight from the registers: U(cc, rr), 16 lanes = 128 contiguous bytes
            // (a transposed read of Vs would be a 16-way bank conflict at this row length)
            v2u xv;
            __builtin_memcpy(&xv, &x, 8);
            __builtin_amdgcn_raw_buffer_store_b64(
                xv, rU, (offD + (unsigned)cc + (unsigned)rr * (unsigned)ld) * 8u, 0, AUX_SC1);
          }
    } else {
      // the two wavefronts with no part in the factorisation: the coming step's tiles. H(j + 1)
      // normally finishes a few microseconds into this step's factorisation: its flag is looked at
      // once per barrier from the 7th on -- never waited for here: with few workgroups H(j + 1)'s
      // ticket may still be behind tasks that wait for this very X_jj.
      v4u tS[16], tB[16];
      bool issued = false;
#pragma unroll 1
      for (int b = 0; b < 16; ++b) {
        if (b >= 6 && !issued && j + 1 < nt &&
            __hip_atomic_load(a.fH + j + 1, MAGMA_RLX_AGENT) != 0) {
          walker_fetch<128>(tid - 128, j + 1, rL, ld, tS, tB);
          issued = true;
        }
        __syncthreads();
      }
      __syncthreads();
      if (issued) walker_park<128>(tid - 128, j + 1, tS, tB, Ws, Ss);
      if (lane == 0) s_walk[wave - 2] = issued ? 1 : 0;
    }
    __syncthreads();
    const bool parked = (s_walk[0] != 0) && (s_walk[1] != 0);
    DAG_STAMP(ts, 3);
    // X_jj (it overwrites the parked S~_jj, which this workgroup has consumed) and U_jj = X_jj^T
    store_tile_sc1(Vs, rL, offD, ld);
    DAG_STAMP(ts, 4);
    publish(a.fL + j * nt + j, a.fU + j * nt + j);
    DAG_STAMP(ts, 5);
    if (j + 1 < nt && !parked) {  // (X_jj is out: whatever H(j + 1) still waits for can now finish)
      if (tid == 0) atomicAdd(a.ctrl + 2, 1);  // diagnostic: steps whose tiles came late
      if (poll_ready(a.fH + j + 1, a.fH + j + 1, 0, 1, a.ctrl, s_ready) < 0) return abandon(a.info);
      v4u tS[8], tB[8];
      walker_fetch<256>(tid, j + 1, rL, ld, tS, tB);
      walker_park<256>(tid, j + 1, tS, tB, Ws, Ss);
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------ walker, second generation
// The same chain, with the two products of a block step taken off it. In dag_walker a step is
//   L_{j,j-1} = B_j X_{j-1}^T  ->  S_jj = S~_jj - L L^T  ->  64 x 64 factor  ->  X_jj,
// all four wavefronts in lock step (the factor's per-block workgroup barriers), two of them idle
// during the factor. But X_jj leaves the SOLVE wavefront four rows at a time, and column c of
// L_{j+1,j} = B_{j+1} X_jj^T needs rows <= c of X_jj only. Here the factor's two wavefronts talk
// through LDS counters instead of barriers (Chol64Flags), SOLVE exports X_jj (to LDS, and U_jj to
// memory) block by block, and the other two wavefronts ("updaters") work one step ahead while the
// factor of step j runs: fetch B_{j+1} and S~_{j+1} when H(j+1) has parked them, then per batch
// of 16 columns -- as soon as those rows of X_jj exist -- the columns of L_{j+1,j} (triangular:
// 4 (k + 1) k-blocks in batch k), their store, and S_{j+1,j+1} -= L_batch L_batch^T (lower tiles,
// in registers). When the factor ends only the last batch is left. One workgroup barrier per
// step; every wait is an LDS (or flag) poll with a bound that raises the abort word.
// LDS: Vs = X_jj (row-major, rows of XLD = 65 doubles: the SOLVE wavefront's lanes hold 16
// consecutive columns of a row, and a column-major store at the operand stride of 80 doubles is an
// 8-way bank conflict that made SOLVE the slower of the two), Ws = B_{j+1} (operand layout), Ss =
// S_jj staged for the factor, Lb[2] = the
// batches of L (the workers' As / Bs), sy = counters (monotonic over the steps: no reset).
constexpr int XLD = 65;
constexpr int WS_PUB = 0, WS_CONS = 1, WS_XROW = 2, WS_ABORT = 3, WS_PARK = 4, WS_LB = 5, WS_XST = 9,
              WS_LST = 10, WS_WORDS = 16;

// this wavefront has arrived; wait until `target` arrivals are in
__device__ __forceinline__ bool lds_arrive_wait(int* ctr, int target, int* abort_) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  return lds_wait_ge(ctr, target, abort_);
}

// a flag another workgroup raises (all lanes of one wavefront poll it)
__device__ __forceinline__ bool wave_wait_flag(const int* flag, int* ctrl, int* abort_) {
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(flag, MAGMA_RLX_AGENT) != 0) return true;
    __builtin_amdgcn_s_sleep(8);
    if ((spins & 255u) == 255u) {
      if (spins >= SPIN_LIMIT || __hip_atomic_load(ctrl + 1, MAGMA_RLX_AGENT) != 0)
        __hip_atomic_store(abort_, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (__hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) return false;
    }
  }
}

__device__ __forceinline__ void dag_walker2(const DagArgs& a, rsrc_t rL, rsrc_t rU, double* Vs,
                                            double* Ws, double* Ss, double* Lb0, double* Lb1,
                                            double* Pn, double* Xch, int* sy, int* s_ready) {
  const int nt = a.nt, ld = a.ld;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = lane & 15, g = lane >> 4;
  int* abort_ = sy + WS_ABORT;
  if (tid < WS_WORDS) sy[tid] = 0;
  {  // step 0's tile (S_00 = A_00, parked by H(0)), by everybody
    if (poll_ready(a.fH, a.fH, 0, 1, a.ctrl, s_ready) < 0) return abandon(a.info);
    v4u tS[8], tB[8];
    walker_fetch<256>(tid, 0, rL, ld, tS, tB);
    walker_park<256>(tid, 0, tS, tB, Ws, Ss);
    __syncthreads();
  }
  for (int j = 0; j < nt; ++j) {
    const unsigned offD = (unsigned)j * NB * (unsigned)(ld + 1);
    const bool next = j + 1 < nt;
    const unsigned offD1 = (unsigned)(j + 1) * NB * (unsigned)(ld + 1);  // S~_{j+1}: tile (j+1, j+1)
    const unsigned offB1 = (unsigned)(j + 1) * NB + (unsigned)j * NB * (unsigned)ld;  // (j+1, j)
    const int ts = nt * nt + nt * (nt + 1) / 2 + j;
    (void)ts;
    DAG_STAMP(ts, 0);
    Chol64Flags fs{sy + WS_PUB, sy + WS_CONS, abort_, 16 * j};
    if (wave == 0) {
      // ---- FACTOR, then its half of the X_jj tile store
      v4d S[4][4];
#pragma unroll
      for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
          for (int v = 0; v < 4; ++v) {
            const int rr = 16 * it + g + 4 * v, cc = 16 * jt + l;
            S[it][jt][v] = (rr >= cc) ? Ss[rr * 65 + cc] : Ss[cc * 65 + rr];
          }
      bool ok = chol64_factor(S, Pn, Xch, lane, fs);
      ok = ok && lds_wait_ge(sy + WS_XROW, 16 * (j + 1), abort_);  // X_jj complete in Vs
      DAG_STAMP(ts, 3);
      if (ok) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int u = lane + i * 128;
          const int m2 = (u & 31) * 2, n = u >> 5;
          double2 xp;
          xp.x = Vs[m2 * XLD + n];
          xp.y = Vs[(m2 + 1) * XLD + n];
          v4u v;
          __builtin_memcpy(&v, &xp, 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, rL, (offD + m2 + (unsigned)n * (unsigned)ld) * 8u, 0, AUX_SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lds_arrive_wait(sy + WS_XST, 2 * (j + 1), abort_) && lane == 0) {
          __hip_atomic_store(a.fL + j * nt + j, 1, MAGMA_RLX_AGENT);
          __hip_atomic_store(a.fU + j * nt + j, 1, MAGMA_RLX_AGENT);
        }
      }
    } else if (wave == 1) {
      // ---- SOLVE: X_jj into Vs and U_jj = X_jj^T into memory as the rows are finished
      v4d B[4][4];
      double pm = 1.0;
      int pe = 0, bad = 0;
      const bool ok = chol64_solve(
          B, Pn + 256, Xch, lane, pm, pe, bad, fs, [&](int b, int jt, double x) {
            const int rr = 4 * b + g, cc = 16 * jt + l;  // X(rr, cc)
            Vs[rr * XLD + cc] = x;
            v2u xv;
            __builtin_memcpy(&xv, &x, 8);
            __builtin_amdgcn_raw_buffer_store_b64(
                xv, rU, (offD + (unsigned)cc + (unsigned)rr * (unsigned)ld) * 8u, 0, AUX_SC1);
            // (announced per batch of 16 rows: the release fence in front of it is a stall)
            if (jt == 3 && (b & 3) == 3) lds_publish(sy + WS_XROW, 16 * j + b + 1);
          });
      if (ok) {
        if (lane == 0) {
          atomicAdd(a.logdet, log(pm) + (double)pe * 0.69314718055994530941723212145818);
          if (bad != 0) atomicCAS(a.info, 0, j * NB + bad);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int u = 64 + lane + i * 128;
          const int m2 = (u & 31) * 2, n = u >> 5;
          double2 xp;
          xp.x = Vs[m2 * XLD + n];
          xp.y = Vs[(m2 + 1) * XLD + n];
          v4u v;
          __builtin_memcpy(&v, &xp, 16);
          __builtin_amdgcn_raw_buffer_store_b128(v, rL, (offD + m2 + (unsigned)n * (unsigned)ld) * 8u, 0, AUX_SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        (void)lds_arrive_wait(sy + WS_XST, 2 * (j + 1), abort_);
      }
    } else if (next) {
      // ---- updaters: step j + 1's S, from the rows of X_jj as they appear
      const int u = wave - 2;       // 0 / 1
      const int q128 = tid - 128;   // index among the updaters' threads
      // lower tiles (mt, nt) of S owned by this wavefront: u = 0: row tiles 0, 3; u = 1: 1, 2
      const int smt[5] = {u ? 1 : 0, u ? 1 : 3, u ? 2 : 3, u ? 2 : 3, u ? 2 : 3};
      const int snt[5] = {0, u ? 1 : 0, u ? 0 : 1, u ? 1 : 2, u ? 2 : 3};
      const int lmt[2] = {u ? 1 : 0, u ? 2 : 3};  // row tiles of L computed by this wavefront
      // S~_{j+1} (HS(j+1): ready early) -> accumulators; then B_{j+1} (HB(j+1)) -> Ws
      bool ok = wave_wait_flag(a.fS + j + 1, a.ctrl, abort_);
      v4d Sa[5];
      if (ok) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const unsigned m = 16 * smt[i] + l, n = 16 * snt[i] + g + 4 * r;
            const v2u w = __builtin_amdgcn_raw_buffer_load_b64(rL, (offD1 + m + n * (unsigned)ld) * 8u, 0, AUX_SC1);
            double d;
            __builtin_memcpy(&d, &w, 8);
            Sa[i][r] = d;
          }
        ok = wave_wait_flag(a.fH + j + 1, a.ctrl, abort_);
      }
      DAG_STAMP_T(ts, 1, 128);
      if (ok) {
        v4u tB[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int e = q128 + i * 128;
          const unsigned off = (unsigned)((e & 31) * 2) + (unsigned)(e >> 5) * (unsigned)ld;
          tB[i] = __builtin_amdgcn_raw_buffer_load_b128(rL, (offB1 + off) * 8u, 0, AUX_SC1);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int e = q128 + i * 128;
          *reinterpret_cast<v4u*>(&Ws[(e >> 5) * TLD + (e & 31) * 2]) = tB[i];
        }
        ok = lds_arrive_wait(sy + WS_PARK, 2 * (j + 1), abort_);
        DAG_STAMP_T(ts, 2, 128);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (!ok) break;
        ok = lds_wait_ge(sy + WS_XROW, 16 * j + 4 * (k + 1), abort_);  // rows < 16 (k + 1) of X_jj
        if (!ok) break;
        double* Lb = (k & 1) ? Lb1 : Lb0;
        v4d Lo[2];
        Lo[0] = (v4d){0.0, 0.0, 0.0, 0.0};
        Lo[1] = (v4d){0.0, 0.0, 0.0, 0.0};
        {  // fragment reads run four k-blocks ahead of the MFMAs (see lds_product64)
          double fb[2][4], fa0[2][4], fa1[2][4];
          auto load = [&](int bt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int kk = 16 * bt + 4 * q + g;
              fb[bt & 1][q] = Vs[(16 * k + l) * XLD + kk];
              fa0[bt & 1][q] = Ws[kk * TLD + 16 * lmt[0] + l];
              fa1[bt & 1][q] = Ws[kk * TLD + 16 * lmt[1] + l];
            }
          };
          load(0);
#pragma unroll
          for (int bt = 0; bt <= k; ++bt) {
            if (bt + 1 <= k) load(bt + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              Lo[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[bt & 1][q], fa0[bt & 1][q], Lo[0], 0, 0, 0);
              Lo[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[bt & 1][q], fa1[bt & 1][q], Lo[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) Lb[(g + 4 * r) * TLD + 16 * lmt[t] + l] = Lo[t][r];
        ok = lds_arrive_wait(sy + WS_LB + k, 2 * (j + 1), abort_);
        if (!ok) break;
        // the batch's 16 columns of L_{j+1,j} to memory
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int e = q128 + i * 128;
          const int m2 = (e & 31) * 2, kk = e >> 5;
          const v4u v = *reinterpret_cast<const v4u*>(&Lb[kk * TLD + m2]);
          __builtin_amdgcn_raw_buffer_store_b128(
              v, rL, (offB1 + m2 + (unsigned)(16 * k + kk) * (unsigned)ld) * 8u, 0, AUX_SC1);
        }
        // S -= L_batch L_batch^T on this wavefront's lower tiles (all fragments read up front)
        {
          double fm[4][5], fn[4][5];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 5; ++i) {
              fm[q][i] = -Lb[(4 * q + g) * TLD + 16 * smt[i] + l];
              fn[q][i] = Lb[(4 * q + g) * TLD + 16 * snt[i] + l];
            }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 5; ++i)
              Sa[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(fn[q][i], fm[q][i], Sa[i], 0, 0, 0);
        }
        if (k == 2) DAG_STAMP_T(ts, 4, 128);
        if (k == 3) {
          // the tile L_{j+1,j} is complete: its flag goes up now (the stores drain under the MFMAs
          // above), H(j + 2) and the F tasks of column j wait for it
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lds_arrive_wait(sy + WS_LST, 2 * (j + 1), abort_) && tid == 128)
            __hip_atomic_store(a.fL + (j + 1) * nt + j, 1, MAGMA_RLX_AGENT);
        }
      }
      if (ok) {
        // S_{j+1,j+1} staged for the next factor (its lower triangle is what is read); the FACTOR
        // wavefront read Ss long ago (at the start of its factor)
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) Ss[(16 * smt[i] + l) * 65 + 16 * snt[i] + g + 4 * r] = Sa[i][r];
        DAG_STAMP_T(ts, 5, 128);
      }
    }
    __syncthreads();
    if (__hip_atomic_load(abort_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) {
      if (tid == 0) __hip_atomic_store(a.ctrl + 1, 1, MAGMA_RLX_AGENT);
      return abandon(a.info);
    }
  }
}

constexpr int DAG_LDS_V = 64 * TLD;  // staged tile (V / result / the diagonal block's S and X)

// L_{i,i-2} = V X_{i-2,i-2}^T formed by the calling task for its own use (i >= 2; the product of
// F(i,i-2), bit for bit), left in LDS as Vs[k * TLD + m] = L(m, k). V = A_{i,i-2} - sum_{k<i-2} L_ik
// L_{i-2,k}^T is complete a block step before X_{i-2,i-2} exists: F(i,i-2) parks it in the (unused)
// lower tile (i, i-2) of U behind fV[i] on its way, and the tasks HB(i) and HS(i) fetch it ahead
// of time. The caller passes a barrier (or a poll_ready) before it reads Vs.
__device__ __forceinline__ bool own_tile_left2(const DagArgs& a, rsrc_t rL, rsrc_t rU, int i, double* Vs,
                                               double* Bs, double* Ss, int* s_ready) {
  const int nt = a.nt, ld = a.ld, c = i - 2;
  if (poll_ready(a.fV + i, a.fV + i, 0, 1, a.ctrl, s_ready) < 0) return false;
  {
    const int tid = threadIdx.x;
    const int m2 = (tid & 31) * 2;
    const unsigned off = (unsigned)i * NB + (unsigned)c * NB * (unsigned)ld;
    v4u v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int n = q * 8 + (tid >> 5);
      v[q] = __builtin_amdgcn_raw_buffer_load_b128(rU, (off + m2 + (unsigned)n * (unsigned)ld) * 8u, 0, AUX_SC1);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int n = q * 8 + (tid >> 5);
      *reinterpret_cast<v4u*>(&Vs[n * TLD + m2]) = v[q];
    }
  }
  TileAcc out;
  tile_zero(out);
  if (!times_xt(Vs, rL, (unsigned)c * NB * (unsigned)(ld + 1), a.fL + c * nt + c, ld, out, Bs, Ss, a.ctrl,
                s_ready, nullptr))
    return false;
  __syncthreads();  // every wavefront has read Vs
  tile_foreach(out, [&](int m, int n, double v) { Vs[n * TLD + m] = v; });
  return true;
}

__global__ __launch_bounds__(256) void spd_inverse_dag_kernel(DagArgs a) {
  __shared__ __attribute__((aligned(16))) double Vs[DAG_LDS_V];
  __shared__ __attribute__((aligned(16))) double Ws[DAG_LDS_V];
  __shared__ __attribute__((aligned(16))) double Ss[64 * 65];
  __shared__ int s_walk[2];
  __shared__ int s_sync[WS_WORDS];
  __shared__ __attribute__((aligned(16))) double As[16 * TLD];
  __shared__ __attribute__((aligned(16))) double Bs[16 * TLD];
  __shared__ __attribute__((aligned(16))) double Pn[512];
  __shared__ __attribute__((aligned(16))) double Xch[CHOL64_XCH];
  __shared__ int s_ticket;
  __shared__ int s_ready;
  static_assert(64 * 65 <= DAG_LDS_V, "diagonal block staging must fit");
  const int nt = a.nt, ld = a.ld;
  const unsigned bytes = (unsigned)ld * (unsigned)ld * 8u;
  const rsrc_t rL = make_rsrc(a.L, bytes), rU = make_rsrc(a.U, bytes);
  const int ntri = nt * (nt + 1) / 2;
  const int total = nt * nt + ntri;
  const int tid = threadIdx.x;
  if (a.walker) {
    // the FIRST workgroup of this launch that starts running walks the chain (claimed with an
    // atomic, not by block index: the order in which workgroups reach a compute unit is up to
    // the dispatcher and to whatever else occupies the chip -- with several of these kernels in
    // flight, workers of one could otherwise fill the compute units the other's walker needs,
    // and the other way round). A worker therefore never runs before its walker does.
    if (tid == 0) s_ticket = atomicCAS(a.ctrl + 4, 0, 1);
    __syncthreads();
    const int claimed = s_ticket;
    __syncthreads();
    if (claimed == 0) {
      if (a.walker == 2) return dag_walker2(a, rL, rU, Vs, Ws, Ss, As, Bs, Pn, Xch, s_sync, &s_ready);
      return dag_walker(a, rL, rU, Vs, Ws, Ss, Pn, Xch, &s_ready, s_walk);
    }
  }
  for (;;) {
    if (tid == 0) {
      int t = atomicAdd(a.ctrl, 1);
      if (__hip_atomic_load(a.ctrl + 1, MAGMA_RLX_AGENT) != 0) t = total;
      s_ticket = t;
    }
    __syncthreads();
    const int t = s_ticket;
    __syncthreads();
    if (t >= total) return;
    DAG_STAMP(t, 0);
    TileAcc acc;
    tile_zero(acc);
    if (t >= nt * nt) {
      // ---- I(i,j): rows i <= nt-1 in order, so the longest sums start first
      const int p = t - nt * nt;
      int i = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
      while (i * (i + 1) / 2 > p) --i;
      while ((i + 1) * (i + 2) / 2 <= p) ++i;
      const int j = p - i * (i + 1) / 2;
      if (!acc_rows(rU, (unsigned)i * NB, a.fU + i * nt, rU, (unsigned)j * NB, a.fU + j * nt, i, nt,
                    ld, acc, As, Bs, Ss, a.ctrl, &s_ready))
        return abandon(a.info);
      double* C = a.Aout + (int64_t)i * NB + (int64_t)j * NB * ld;
      double* Ct = a.Aout + (int64_t)j * NB + (int64_t)i * NB * ld;
      const bool diag = (i == j);
      DAG_STAMP(t, 1);
      tile_foreach(acc, [&](int m, int n, double v) {
        C[m + (int64_t)n * ld] = v;
        if (!diag) Ct[n + (int64_t)m * ld] = v;
      });
      DAG_STAMP(t, 4);
      continue;
    }
    const int j = t / nt, r = t - j * nt;
    if (r == 0 && a.walker == 2 && j > 0) {
      // ---- HB(j) (walker 2): B_j = A_{j,j-1} - sum_{k<j-1} L_jk L_{j-1,k}^T alone. Its last term
      // needs the walker's own L_{j-1,j-2}, so it sits on the walker's critical cycle: S~_jj, which
      // does not, has a task of its own (HS below) and half the MFMAs are gone from this one.
      // That last term's other factor, L_{j,j-2}, used to arrive through memory from F(j,j-2): X
      // seen -> product -> store -> flag -> poll here -> load -> product, 8-9 us after X_{j-2,j-2}
      // with the walker's updaters waiting for B_j half of every step. Now this task forms
      // L_{j,j-2} itself (the same sums and the same product as F(j,j-2), which still publishes the
      // tile for everybody else) and takes the term from LDS: one hand-off less on the cycle.
      tile_load_neg(acc, a.A + (int64_t)j * NB + (int64_t)(j - 1) * NB * ld, ld);
      if (j >= 2) {
        if (!acc_rows(rL, (unsigned)j * NB, a.fL + j * nt, rL, (unsigned)(j - 1) * NB, a.fL + (j - 1) * nt,
                      0, j - 2, ld, acc, As, Bs, Ss, a.ctrl, &s_ready))
          return abandon(a.info);
        if (!own_tile_left2(a, rL, rU, j, Vs, Bs, Ss, &s_ready)) return abandon(a.info);
        // acc += L_{j,j-2} (LDS) L_{j-1,j-2}^T (the walker's, behind its flag)
        if (!times_xt(Vs, rL, (unsigned)(j - 1) * NB + (unsigned)(j - 2) * NB * (unsigned)ld,
                      a.fL + (j - 1) * nt + (j - 2), ld, acc, Bs, Ss, a.ctrl, &s_ready, nullptr))
          return abandon(a.info);
      }
      __syncthreads();
      tile_foreach(acc, [&](int m, int n, double v) { Ws[n * TLD + m] = -v; });
      __syncthreads();
      store_tile_sc1(Ws, rL, (unsigned)j * NB + (unsigned)(j - 1) * NB * (unsigned)ld, ld);
      publish(a.fH + j, nullptr);
      __syncthreads();
      continue;
    }
    if (r == 1 && a.walker == 2 && j + 1 < nt) {
      // ---- HS(j+1) (walker 2, in the slot of F(j+1,j), which the walker computes):
      // S~_{j+1,j+1} = A_{j+1,j+1} - sum_{k<j} L_{j+1,k} L_{j+1,k}^T, parked in tile (j+1, j+1). Its
      // last term, k = j-1, from this task's own copy of L_{j+1,j-1} in LDS (see HB above).
      const int i = j + 1;
      tile_load_neg(acc, a.A + (int64_t)i * NB * (ld + 1), ld);
      if (i >= 2) {
        if (!acc_rows(rL, (unsigned)i * NB, a.fL + i * nt, rL, (unsigned)i * NB, a.fL + i * nt, 0, i - 2,
                      ld, acc, As, Bs, Ss, a.ctrl, &s_ready))
          return abandon(a.info);
        if (!own_tile_left2(a, rL, rU, i, Vs, Bs, Ss, &s_ready)) return abandon(a.info);
        __syncthreads();  // L_{i,i-2} staged by every wavefront
        lds_product64(Vs, Vs, acc);
      }
      __syncthreads();
      tile_foreach(acc, [&](int m, int n, double v) { Vs[n * TLD + m] = -v; });
      __syncthreads();
      store_tile_sc1(Vs, rL, (unsigned)i * NB * (unsigned)(ld + 1), ld);
      publish(a.fS + i, nullptr);
      __syncthreads();
      continue;
    }
    if (r == 0 && a.walker) {
      // ---- H(j): what block step j of the walker needs from the other tiles, one step ahead
      tile_load_neg(acc, a.A + (int64_t)j * NB * (ld + 1), ld);
      TileAcc accB;
      tile_zero(accB);
      if (j > 0) {
        tile_load_neg(accB, a.A + (int64_t)j * NB + (int64_t)(j - 1) * NB * ld, ld);
        if (!acc_rows_t<true>(rL, (unsigned)j * NB, a.fL + j * nt, rL, (unsigned)(j - 1) * NB,
                              a.fL + (j - 1) * nt, 0, j - 1, ld, accB, acc, As, Bs, Ss, a.ctrl,
                              &s_ready))
          return abandon(a.info);
      }
      __syncthreads();
      tile_foreach(acc, [&](int m, int n, double v) { Vs[n * TLD + m] = -v; });
      if (j > 0) tile_foreach(accB, [&](int m, int n, double v) { Ws[n * TLD + m] = -v; });
      __syncthreads();
      store_tile_sc1(Vs, rL, (unsigned)j * NB * (unsigned)(ld + 1), ld);
      if (j > 0) store_tile_sc1(Ws, rL, (unsigned)j * NB + (unsigned)(j - 1) * NB * (unsigned)ld, ld);
      publish(a.fH + j, nullptr);
      __syncthreads();
      continue;
    }
    if (r == 0) {
      // ---- D(j): the tile left of the diagonal, L_{j,j-1}, then F(j,j), by ONE workgroup. The
      // chain of diagonal tiles runs through L_{j,j-1} (it is the last term of S_jj); computed
      // here it goes from LDS straight into that term instead of through memory, a flag, a poll
      // and a fetch (4-5 us of the 24 us a block step took). Both tiles have accumulated all
      // their earlier terms long before X_{j-1,j-1} appears.
      tile_load_neg(acc, a.A + (int64_t)j * NB * (ld + 1), ld);
      if (j > 0) {
        // both tiles take their terms k < j-1 from the same slabs of block row j, as they appear
        TileAcc accB;
        tile_load_neg(accB, a.A + (int64_t)j * NB + (int64_t)(j - 1) * NB * ld, ld);
        if (!acc_rows_t<true>(rL, (unsigned)j * NB, a.fL + j * nt, rL, (unsigned)(j - 1) * NB,
                              a.fL + (j - 1) * nt, 0, j - 1, ld, accB, acc, As, Bs, Ss, a.ctrl,
                              &s_ready))
          return abandon(a.info);
        // parked in LDS as the operand of the product with X^T
        __syncthreads();
        tile_foreach(accB, [&](int m, int n, double v) { Vs[n * TLD + m] = -v; });
      }
      if (j > 0) {
        TileAcc out;
        tile_zero(out);
        if (!times_xt(Vs, rL, (unsigned)(j - 1) * NB * (unsigned)(ld + 1),
                      a.fL + (j - 1) * nt + (j - 1), ld, out, Bs, Ss, a.ctrl, &s_ready, nullptr))
          return abandon(a.info);
        __syncthreads();  // every wavefront has read Vs
        tile_foreach(out, [&](int m, int n, double v) { Vs[n * TLD + m] = v; });
        __syncthreads();
        store_tile_sc1(Vs, rL, (unsigned)j * NB + (unsigned)(j - 1) * NB * (unsigned)ld, ld);
        // acc += L_{j,j-1} L_{j,j-1}^T straight from LDS (Vs[k * TLD + m] = L(m, k))
        lds_product64(Vs, Vs, acc);
      }
      DAG_STAMP(t, 1);
      __syncthreads();  // (also ends the reads of Vs above)
      tile_foreach(acc, [&](int m, int n, double v) { Vs[m * 65 + n] = -v; });
      // L_{j,j-1}'s flag goes up here, its stores having drained under the staging of S: nobody
      // on the chain of diagonal tiles waits for it
      if (j > 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (j > 0 && tid == 0) __hip_atomic_store(a.fL + j * nt + (j - 1), 1, MAGMA_RLX_AGENT);
      DAG_STAMP(t, 2);
      const int lane = tid & 63, wave = tid >> 6;
      const int l = lane & 15, g = lane >> 4;
      if (wave == 0) {
        v4d S[4][4];
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int rr = 16 * it + g + 4 * v, cc = 16 * jt + l;
              S[it][jt][v] = (rr >= cc) ? Vs[rr * 65 + cc] : Vs[cc * 65 + rr];
            }
        chol64_factor(S, Pn, Xch, lane);
        __syncthreads();  // S was read before the factor's barriers; X may now overwrite it
      } else if (wave == 1) {
        v4d B[4][4];
        double pm = 1.0;
        int pe = 0, bad = 0;
        chol64_solve(B, Pn + 256, Xch, lane, pm, pe, bad);
        if (lane == 0) {
          // steps are ordered by the flag chain, so the adds arrive in step order
          atomicAdd(a.logdet, log(pm) + (double)pe * 0.69314718055994530941723212145818);
          if (bad != 0) atomicCAS(a.info, 0, j * NB + bad);
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int v = 0; v < 4; ++v) {
              const int rr = 16 * it + g + 4 * v, cc = 16 * jt + l;
              Vs[cc * TLD + rr] = (jt <= it) ? B[it][jt][v] : 0.0;  // X(rr, cc), column cc
            }
      } else {
        chol64_idle();
        __syncthreads();
      }
      __syncthreads();
      DAG_STAMP(t, 3);
      // X_jj -> diagonal tile of L (columns contiguous); U_jj = X_jj^T -> diagonal tile of U
      // X_jj first, with its own flag: the next block step waits for X only
      store_tile_sc1(Vs, rL, (unsigned)j * NB * (unsigned)(ld + 1), ld);
      publish(a.fL + j * nt + j, nullptr);
      {
        const int m2 = (tid & 31) * 2;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int n = c * 8 + (tid >> 5);  // U(m, n) = X(n, m)
          double2 u;
          u.x = Vs[m2 * TLD + n];
          u.y = Vs[(m2 + 1) * TLD + n];
          v4u v;
          __builtin_memcpy(&v, &u, 16);
          __builtin_amdgcn_raw_buffer_store_b128(
              v, rU, ((unsigned)j * NB * (unsigned)(ld + 1) + m2 + (unsigned)n * (unsigned)ld) * 8u,
              0, AUX_SC1);
        }
      }
      publish(a.fU + j * nt + j, nullptr);
      DAG_STAMP(t, 4);
      __syncthreads();
      continue;
    }
    const bool isF = r < nt - j;
    if (isF && r == 1) continue;  // L_{j+1,j} belongs to D(j+1)
    unsigned offOut;
    rsrc_t rOut;
    int* flag;
    if (isF) {
      // ---- F(i,j), i > j
      const int i = j + r;
      tile_load_neg(acc, a.A + (int64_t)i * NB + (int64_t)j * NB * ld, ld);
      if (!acc_rows(rL, (unsigned)i * NB, a.fL + i * nt, rL, (unsigned)j * NB, a.fL + j * nt, 0, j,
                    ld, acc, As, Bs, Ss, a.ctrl, &s_ready))
        return abandon(a.info);
      __syncthreads();
      tile_foreach(acc, [&](int m, int n, double v) { Vs[n * TLD + m] = -v; });
      offOut = (unsigned)i * NB + (unsigned)j * NB * (unsigned)ld;
      rOut = rL;
      flag = a.fL + i * nt + j;
      if (a.walker == 2 && r == 2) {
        // V_{i,i-2}, complete a block step before X_jj: parked for HB(i) and HS(i) (own_tile_left2)
        __syncthreads();
        store_tile_sc1(Vs, rU, offOut, ld);
        publish(a.fV + i, nullptr);
      }
    } else {
      // ---- T(c,j), c < j
      const int c = r - (nt - j);
      if (!acc_rows(rU, (unsigned)c * NB, a.fU + c * nt, rL, (unsigned)j * NB, a.fL + j * nt, c, j,
                    ld, acc, As, Bs, Ss, a.ctrl, &s_ready))
        return abandon(a.info);
      __syncthreads();
      tile_foreach(acc, [&](int m, int n, double v) { Vs[n * TLD + m] = -v; });
      offOut = (unsigned)c * NB + (unsigned)j * NB * (unsigned)ld;
      rOut = rU;
      flag = a.fU + c * nt + j;
    }
    DAG_STAMP(t, 1);
    TileAcc out;
    tile_zero(out);
    if (!times_xt(Vs, rL, (unsigned)j * NB * (unsigned)(ld + 1), a.fL + j * nt + j, ld, out, Bs, Ss,
                  a.ctrl, &s_ready, a.stamps ? a.stamps + t * 8 + 2 : nullptr))
      return abandon(a.info);
    DAG_STAMP(t, 3);
    __syncthreads();  // every wavefront has read Vs
    tile_foreach(out, [&](int m, int n, double v) { Vs[n * TLD + m] = v; });
    __syncthreads();
    store_tile_sc1(Vs, rOut, offOut, ld);
    publish(flag, nullptr);
    DAG_STAMP(t, 4);
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void dag_init_kernel(int* __restrict__ ctrl, int nwords,
                                                        double* __restrict__ logdet,
                                                        int* __restrict__ info) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < nwords) ctrl[k] = 0;
  if (k == 0) {
    *logdet = 0.0;
    *info = 0;
  }
}

size_t spd_inverse_dag_ctrl_bytes(int nt) {
  return (((size_t)(8 + 2 * nt * nt + 3 * nt) * sizeof(int)) + 255) & ~(size_t)255;
}

static int g_dag_workgroups = 256;
static thread_local int t_dag_grid_once = 0;  // workgroups of this thread's NEXT launch (0 = default)
void set_dag_grid_once(int v) { t_dag_grid_once = v; }
static int g_dag_walker = 2;
void set_dag_walker(int v) { g_dag_walker = v; }
static long long* g_dag_stamps = nullptr;
void set_dag_workgroups(int v) { g_dag_workgroups = v; }
void set_dag_stamps(void* p) { g_dag_stamps = static_cast<long long*>(p); }

// Round 2: the tile graph won while the factorisation was latency bound (N <= 4000) and lost to the
// launch-per-step path's wide launches from ~70 block rows on (N = 8000: 19.8 against 14.9 ms).
// With the pipelined tile products of round 3 (acc_rows_t) it wins at every size measured: N =
// 2000 / 4000 / 6000 / 8000 / 16000: 0.50 / 1.71 / 5.36 / 10.65 / 79.7 ms (launch-per-step:
// 1.15 / 3.4 / 7.4 / 14.7 ms). 256 block rows = N <= 16384.
static int g_dag_max_block_rows = 256;
void set_dag_max_block_rows(int v) { g_dag_max_block_rows = v; }

bool spd_inverse_dag_supported(int Np) {
  return (size_t)Np * Np * 8 < ((size_t)1 << 32) && g_dag_workgroups > 0 &&
         Np / NB <= g_dag_max_block_rows;
}

// work: 3 Np^2 doubles (L, U, control block at the head of the third).
int launch_spd_inverse_dag(int Np, double* A, double* logdet, int* info, double* work,
                           hipStream_t st) {
  const int nt = Np / NB;
  const size_t n2 = (size_t)Np * Np;
  DagArgs a;
  a.A = A;
  a.Aout = A;
  a.L = work;
  a.U = work + n2;
  int* ctrl = reinterpret_cast<int*>(work + 2 * n2);
  a.ctrl = ctrl;
  a.fL = ctrl + 8;
  a.fU = ctrl + 8 + nt * nt;
  a.fH = ctrl + 8 + 2 * nt * nt;
  a.fS = a.fH + nt;
  a.fV = a.fS + nt;
  a.walker = g_dag_walker < 0 ? 0 : (g_dag_walker > 2 ? 2 : g_dag_walker);
  a.logdet = logdet;
  a.info = info;
  a.stamps = g_dag_stamps;
  a.nt = nt;
  a.ld = Np;
  if (spd_inverse_dag_ctrl_bytes(nt) > n2 * sizeof(double))
    return fail(2, "spd_inverse: workspace too small for the control block");
  // control block, log-det and info cleared by one small kernel (three memsets cost three
  // launches in front of every inverse of the serial theta_0 chain)
  const int nwords = (int)(spd_inverse_dag_ctrl_bytes(nt) / sizeof(int));
  hipLaunchKernelGGL(dag_init_kernel, dim3((nwords + 255) / 256), dim3(256), 0, st, ctrl, nwords, logdet, info);
  const int total = nt * nt + nt * (nt + 1) / 2;
  // (walker mode: one more workgroup -- whichever starts first walks the diagonal chain)
  const int want = t_dag_grid_once > 0 ? t_dag_grid_once : g_dag_workgroups;
  t_dag_grid_once = 0;
  const int grid = (total < want ? total : want) + (a.walker ? 1 : 0);
  hipLaunchKernelGGL(spd_inverse_dag_kernel, dim3(grid), dim3(256), 0, st, a);
  return check_hip(hipGetLastError(), "spd_inverse_dag");
}

}  // namespace magma
