// probe of v_mfma_f64_16x16x4_f64 operand / result layout on gfx950 (developer tool)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double* A /*16x4 row-major*/, const double* B /*4x16 row-major*/, double* out /*64 lanes x 4*/) {
  const int l = threadIdx.x;
  const double a = A[(l % 16) * 4 + (l / 16)];   // hypothesis: A[i][k], i = l%16, k = l/16
  const double b = B[(l / 16) * 16 + (l % 16)];  // hypothesis: B[k][j], j = l%16, k = l/16
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  double hA[64], hB[64], hD[256], ref[16][16];
  srand(1);
  for (int i = 0; i < 64; ++i) { hA[i] = (rand() % 17) - 8; hB[i] = (rand() % 13) - 6; }
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i][j] = s; }
  double *dA, *dB, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, hA, 512, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 2048, hipMemcpyDeviceToHost);
  int ok1 = 1, ok2 = 1;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    if (hD[l * 4 + r] != ref[4 * (l / 16) + r][l % 16]) ok1 = 0;   // i = 4*(l/16) + r
    if (hD[l * 4 + r] != ref[4 * r + (l / 16)][l % 16]) ok2 = 0;   // i = 4*r + l/16
  }
  printf("layout i=4*(l/16)+r: %s ; layout i=4*r+(l/16): %s\n", ok1 ? "MATCH" : "no", ok2 ? "MATCH" : "no");
  return 0;
}
