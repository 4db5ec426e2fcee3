// GPU probe: empirical lane / byte map of ds_read_b64_tr_b8 (gfx950).  Every lane supplies the address of 8 contiguous bytes; the 512-byte region is
// filled with two patterns (index & 255, index >> 8) so that the source byte index of every output byte can be recovered.  Run via gpurun:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_tr8.hip -o tools/probes/probe_tr8 && tools/probes/probe_tr8 [stride_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v2i __attribute__((ext_vector_type(2)));

__global__ void k_tr8(int pattern, int stride, unsigned char* out) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[64 * 64];
  const int l = threadIdx.x;
  for (int i = l; i < 64 * 64; i += 64) lds[i] = pattern == 0 ? (unsigned char)(i & 255) : (unsigned char)(i >> 8);
  __syncthreads();
  v2i v;
  const unsigned addr = (unsigned)(size_t)(const char __attribute__((address_space(3)))*)(lds + l * stride);
  asm volatile("ds_read_b64_tr_b8 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int e = 0; e < 8; ++e) out[l * 8 + e] = (unsigned char)((e < 4 ? v[0] >> (8 * e) : v[1] >> (8 * (e - 4))) & 255);
}

int main(int argc, char** argv) {
  const int stride = argc > 1 ? atoi(argv[1]) : 8;
  unsigned char* d; hipMalloc(&d, 512);
  std::vector<unsigned char> lo(512), hi(512);
  k_tr8<<<1, 64>>>(0, stride, d); hipMemcpy(lo.data(), d, 512, hipMemcpyDeviceToHost);
  k_tr8<<<1, 64>>>(1, stride, d); hipMemcpy(hi.data(), d, 512, hipMemcpyDeviceToHost);
  printf("ds_read_b64_tr_b8, lane address = base + lane * %d: out[lane][byte] = source byte index (as lane:byte of the supplying lane)\n", stride);
  for (int l = 0; l < 64; ++l) {
    printf("lane %2d:", l);
    for (int e = 0; e < 8; ++e) {
      const int idx = lo[l * 8 + e] | (hi[l * 8 + e] << 8);
      printf("  %2d:%d", idx / stride, idx % stride);
    }
    printf("\n");
  }
  return 0;
}
