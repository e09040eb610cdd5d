// Internal pieces shared by capi.hip and driver.hip (the native EM driver).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace magma {

// Split of the theta_0 gradient's two N^3 products over ranks: rank r forms the
// block rows i = r, r + world, ... of T = K_0^-1 C and the lower tiles (i, j <= i) of Z = T K_0^-1
// in those rows (plist: their triangular indices, device memory), contracts them, and the two
// gradient sums are all-reduced. The inverse and the O(N^2) parts are replicated.
struct Theta0Shard {
  int rank, world;
  const int* plist;
  int nplist;
  int (*allreduce)(void* user, double* dev, int64_t n, hipStream_t st);
  void* user;
};

int theta0_eval_impl(int N, const double* grid, double theta1, double theta2, double pen_diag,
                     const double* mean, const double* m0, const double* cov, int want_grad,
                     void* work, double* h_out3, int* h_info, hipStream_t st, const Theta0Shard* sh);
int theta0_mstep_impl(int N, const double* grid, const double* h_theta0, double pen_diag,
                      const double* mean, const double* m0, const double* cov, int m, double factr,
                      double pgtol, int maxit, void* work, double* h_theta_out, double* h_f_out,
                      double* h_f_x0, int* h_iters, int* h_nfev, int* h_status, hipStream_t st,
                      const Theta0Shard* sh);

// out4 = (sum f, sum g[:,0], sum g[:,1], sum g[:,2]) over M entries, fixed order (g may be null)
int launch_sum4(int64_t M, const double* f, const double* g, double* out4, hipStream_t st);
// lower 64 x 64 tiles of the Np x Np matrix P <-> contiguous buffer (tile (i, j <= i) at
// (i (i + 1) / 2 + j) * 4096); unpack != 0 copies back into P
int launch_pack_lower(int Np, double* P, double* pack, int unpack, hipStream_t st);

}  // namespace magma
