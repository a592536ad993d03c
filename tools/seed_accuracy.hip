// Measures the relative error of v_rcp_f64 / v_rsq_f64 (and f32) seeds and after 1 / 2 Newton steps on gfx950.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(const double* x, double* out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double y0 = __builtin_amdgcn_rcp(v);
  double e = __builtin_fma(-v, y0, 1.0);
  double y1 = __builtin_fma(y0, e, y0);
  e = __builtin_fma(-v, y1, 1.0);
  double y2 = __builtin_fma(y1, e, y1);
  double r0 = __builtin_amdgcn_rsq(v);
  e = __builtin_fma(-v * r0, r0, 1.0);
  double r1 = __builtin_fma(r0, 0.5 * e, r0);
  e = __builtin_fma(-v * r1, r1, 1.0);
  double r2 = __builtin_fma(r1, 0.5 * e, r1);
  float vf = (float)v;
  float f0 = __builtin_amdgcn_rcpf(vf), g0 = __builtin_amdgcn_rsqf(vf);
  out[i * 8 + 0] = y0; out[i * 8 + 1] = y1; out[i * 8 + 2] = y2;
  out[i * 8 + 3] = r0; out[i * 8 + 4] = r1; out[i * 8 + 5] = r2;
  out[i * 8 + 6] = f0; out[i * 8 + 7] = g0;
}
int main() {
  const int n = 1 << 20;
  double* hx = new double[n];
  for (int i = 0; i < n; ++i) hx[i] = std::exp(-14.0 + 28.0 * (i + 0.37) / n);
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 64);
  hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  double* ho = new double[n * 8];
  hipMemcpy(ho, dout, n * 64, hipMemcpyDeviceToHost);
  double m[8] = {0};
  for (int i = 0; i < n; ++i) {
    double rc = 1.0 / hx[i], rs = 1.0 / std::sqrt(hx[i]);
    float xf = (float)hx[i];
    double rcf = 1.0 / (double)xf, rsf = 1.0 / std::sqrt((double)xf);
    double ref[8] = {rc, rc, rc, rs, rs, rs, rcf, rsf};
    for (int c = 0; c < 8; ++c) m[c] = std::fmax(m[c], std::fabs(ho[i * 8 + c] / ref[c] - 1.0));
  }
  printf("rcp_f64 seed %.3e  1NR %.3e  2NR %.3e\n", m[0], m[1], m[2]);
  printf("rsq_f64 seed %.3e  1NR %.3e  2NR %.3e\n", m[3], m[4], m[5]);
  printf("rcp_f32 seed %.3e  rsq_f32 seed %.3e\n", m[6], m[7]);
  return 0;
}
