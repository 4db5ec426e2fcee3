// GPU probe: empirical lane maps for ds_read_b64_tr_b16 and the bf16 MFMA fragments (run via gpurun).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void k_tr(int* out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * 4];
  int l = threadIdx.x;
  // lane a's 4 elements hold value a*4+e
  for (int e = 0; e < 4; ++e) lds[l * 4 + e] = (short)(l * 4 + e);
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + l * 4));
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}

__device__ __bf16 tobf(float f) { return (__bf16)f; }

// A[i][k] = (i==k) identity-ish probes: compute D = A*B with A one-hot to recover maps.
__global__ void k_mfma16(const float* A /*16x32*/, const float* B /*32x16 as [k][n]*/, float* D /*16x16*/) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    int k = (l >> 4) * 8 + j;
    a[j] = tobf(A[(l & 15) * 32 + k]);
    b[j] = tobf(B[k * 16 + (l & 15)]);
  }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
__global__ void k_mfma32(const float* A /*32x16*/, const float* B /*16x32 [k][n]*/, float* D /*32x32*/) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    int k = (l >> 5) * 8 + j;
    a[j] = tobf(A[(l & 31) * 16 + k]);
    b[j] = tobf(B[k * 32 + (l & 31)]);
  }
  f32x16 c = {};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

int main() {
  int* d; hipMalloc(&d, 256 * 4);
  k_tr<<<1, 64>>>(d);
  std::vector<int> h(256); hipMemcpy(h.data(), d, 1024, hipMemcpyDeviceToHost);
  printf("TR: out[lane][e] = src lane*4+e value\n");
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 4; ++e) {
      printf(" %3d", h[l * 4 + e]);
      int g = l >> 4, i = l & 15;
      int expect = (g * 16 + 4 * e + (i >> 2)) * 4 + (i & 3);
      if (h[l * 4 + e] != expect) ok = 0;
    }
    printf("\n");
  }
  printf("TR hypothesis out[i][j]=in[4j+(i>>2)][i&3] per 16-lane group: %s\n", ok ? "CONFIRMED" : "WRONG");

  // MFMA checks with asymmetric random-ish integer matrices
  {
    std::vector<float> A(16 * 32), B(32 * 16), D(256), R(256, 0.f);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) A[i * 32 + k] = (float)((i * 7 + k * 3) % 5 - 2);
    for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) B[k * 16 + n] = (float)((k * 5 + n * 11) % 7 - 3);
    for (int i = 0; i < 16; ++i) for (int n = 0; n < 16; ++n) for (int k = 0; k < 32; ++k) R[i * 16 + n] += A[i * 32 + k] * B[k * 16 + n];
    float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k_mfma16<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; ++i) if (D[i] != R[i]) ++bad;
    printf("MFMA 16x16x32 bf16 layout: %s (%d mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
  }
  {
    std::vector<float> A(32 * 16), B(16 * 32), D(1024), R(1024, 0.f);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 7 + k * 3) % 5 - 2);
    for (int k = 0; k < 16; ++k) for (int n = 0; n < 32; ++n) B[k * 32 + n] = (float)((k * 5 + n * 11) % 7 - 3);
    for (int i = 0; i < 32; ++i) for (int n = 0; n < 32; ++n) for (int k = 0; k < 16; ++k) R[i * 32 + n] += A[i * 16 + k] * B[k * 32 + n];
    float *dA, *dB, *dD; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k_mfma32<<<1, 64>>>(dA, dB, dD); hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 1024; ++i) if (D[i] != R[i]) ++bad;
    printf("MFMA 32x32x16 bf16 layout: %s (%d mismatches)\n", bad ? "WRONG" : "CONFIRMED", bad);
  }
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device: %s CUs=%d clock=%d kHz lds/block=%zu\n", p.name, p.multiProcessorCount, p.clockRate, p.sharedMemPerBlock);
  return 0;
}
