// Shared declarations for libmagma_hip.so (gfx950 / MI355X only, FP64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#define MAGMA_NSMALL_MAX 64  // one wavefront per individual: n_i <= 64

namespace magma {

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
