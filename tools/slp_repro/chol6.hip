// Stand-alone reproducer candidate for the SLP miscompile of rigid_step4_kernel (tools/slp_repro/README.md): the root's 6x6 SPD solve
// (Cholesky + two substitutions, the statements of add-gym_amd/csrc/rigid.hip) on one system per lane, checked against a double-precision
// host solve.  Build twice:  hipcc -O3 --offload-arch=gfx950 chol6.hip -o chol6_slp ;  ... -fno-slp-vectorize ... -o chol6_noslp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void solve6(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i >= n) return;
  const float* s = in + (size_t)i * 27;  // A xx xy xz yy yz zz | B[9] | C xx xy xz yy yz zz | p.a xyz | p.l xyz   (ArtI + bias force)
  float A[6][6];
  A[0][0] = s[0]; A[0][1] = s[1]; A[0][2] = s[2]; A[1][1] = s[3]; A[1][2] = s[4]; A[2][2] = s[5];
  A[3][3] = s[15]; A[3][4] = s[16]; A[3][5] = s[17]; A[4][4] = s[18]; A[4][5] = s[19]; A[5][5] = s[20];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) A[a][3 + b] = s[6 + 3 * a + b];
  float b[6] = {-s[21], -s[22], -s[23], -s[24], -s[25], -s[26]};
#pragma unroll
  for (int a = 0; a < 6; ++a) {
#pragma unroll
    for (int j = a; j < 6; ++j) {
      float sum = A[a][j];
#pragma unroll
      for (int t = 0; t < a; ++t) sum -= A[t][a] * A[t][j];
      A[a][j] = (j == a) ? sqrtf(fmaxf(sum, 1e-20f)) : sum / A[a][a];
    }
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) {
    float sum = b[a];
#pragma unroll
    for (int t = 0; t < a; ++t) sum -= A[t][a] * b[t];
    b[a] = sum / A[a][a];
  }
#pragma unroll
  for (int a = 5; a >= 0; --a) {
    float sum = b[a];
#pragma unroll
    for (int t = a + 1; t < 6; ++t) sum -= A[a][t] * b[t];
    b[a] = sum / A[a][a];
  }
#pragma unroll
  for (int a = 0; a < 6; ++a) out[(size_t)i * 6 + a] = b[a];
}

int main() {
  const int n = 4096;
  std::vector<float> in((size_t)n * 27), out((size_t)n * 6);
  unsigned long long st = 12345;
  auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (float)((st >> 33) & 0xffffff) / 8388608.0f - 1.0f; };
  std::vector<double> ref((size_t)n * 6);
  for (int i = 0; i < n; ++i) {
    double G[6][6], M[6][6], rhs[6];
    for (auto& r : G) for (auto& x : r) x = rnd();
    for (int a = 0; a < 6; ++a)
      for (int c = 0; c < 6; ++c) { M[a][c] = (a == c) ? 3.0 : 0.0; for (int k = 0; k < 6; ++k) M[a][c] += G[a][k] * G[c][k]; }
    float Mf[6][6];  // the system as the kernel sees it: fp32, symmetric
    for (int a = 0; a < 6; ++a) for (int c = 0; c < 6; ++c) Mf[a][c] = (float)M[a < c ? a : c][a < c ? c : a];
    float* s = &in[(size_t)i * 27];
    const int sym[6][2] = {{0, 0}, {0, 1}, {0, 2}, {1, 1}, {1, 2}, {2, 2}};
    for (int k = 0; k < 6; ++k) { s[k] = Mf[sym[k][0]][sym[k][1]]; s[15 + k] = Mf[3 + sym[k][0]][3 + sym[k][1]]; }
    for (int a = 0; a < 3; ++a) for (int c = 0; c < 3; ++c) s[6 + 3 * a + c] = Mf[a][3 + c];
    for (int k = 0; k < 6; ++k) { s[21 + k] = rnd() * 5.0f; rhs[k] = -(double)s[21 + k]; }
    double T[6][7];  // solved in double by Gaussian elimination
    for (int a = 0; a < 6; ++a) { for (int c = 0; c < 6; ++c) T[a][c] = Mf[a][c]; T[a][6] = rhs[a]; }
    for (int p = 0; p < 6; ++p) for (int r = p + 1; r < 6; ++r) { const double f = T[r][p] / T[p][p]; for (int c = p; c < 7; ++c) T[r][c] -= f * T[p][c]; }
    for (int p = 5; p >= 0; --p) { double v = T[p][6]; for (int c = p + 1; c < 6; ++c) v -= T[p][c] * ref[(size_t)i * 6 + c]; ref[(size_t)i * 6 + p] = v / T[p][p]; }
  }
  float *din, *dout;
  if (hipMalloc(&din, in.size() * 4) != hipSuccess || hipMalloc(&dout, out.size() * 4) != hipSuccess) return 2;
  (void)hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(solve6, dim3(n / 64), dim3(64), 0, 0, din, dout, n);
  if (hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  double worst = 0;
  for (size_t k = 0; k < out.size(); ++k) worst = fmax(worst, fabs(out[k] - ref[k]) / fmax(1.0, fabs(ref[k])));
  printf("chol6: worst error %.3e of scale -> %s\n", worst, worst < 1e-4 ? "ok" : "WRONG");
  return worst < 1e-4 ? 0 : 1;
}
