// Shared declarations for libmagma_hip.so (gfx950 / MI355X only, FP64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#define MAGMA_NSMALL_MAX 64  // one wavefront per individual: n_i <= 64

namespace magma {

// Per-individual kernels are instantiated for these row counts (see MAGMA_DISPATCH_NMAX).
constexpr int small_template_nmax(int nmax) {
  return nmax <= 8 ? 8 : nmax <= 16 ? 16 : nmax <= 24 ? 24 : nmax <= 32 ? 32 : nmax <= 40 ? 40
       : nmax <= 48 ? 48 : nmax <= 52 ? 52 : nmax <= 56 ? 56 : 64;
}
// Packed C_i (magma_indiv_gather_cov): ldp rows x small_template_nmax columns per individual.
constexpr int cpack_ldp(int nmax) { return 16 * ((small_template_nmax(nmax) + 15) / 16); }

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);
int check_hip(hipError_t e, const char* what);

// SE-kernel hyper-parameters unpacked once per wavefront.
struct SEParams {
  double th1;   // theta1
  double h;     // exp(-theta2)/2
  double sig2;  // sigma^2
  double eth1;  // exp(theta1)
};

__device__ __forceinline__ SEParams se_params(double th1, double th2, double sigma) {
  SEParams p;
  p.th1 = th1;
  p.h = 0.5 * exp(-th2);
  p.sig2 = sigma * sigma;
  p.eth1 = exp(th1);
  return p;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

}  // namespace magma

#define MAGMA_CHECK_HIP(expr)                                  \
  do {                                                         \
    int _st = magma::check_hip((expr), #expr);                 \
    if (_st) return _st;                                       \
  } while (0)
