// gfc_erff (common.h) against the device library's erff, bit for bit over ALL 2^32 float inputs.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Iinclude tools/micro/erf_check.hip -o /tmp/erf_check && /tmp/erf_check
#include "../../glue-factory-colon_amd/csrc/common.h"

#include <cstdio>

__global__ void check(unsigned long long* bad, unsigned* first) {
  const unsigned long long n = 1ull << 32;
  unsigned long long mism = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (unsigned long long)gridDim.x * blockDim.x) {
    const float x = __uint_as_float((unsigned)i);
    const unsigned a = __float_as_uint(gfc_erff(x)), b = __float_as_uint(erff(x));
    const bool nan_both = (a & 0x7fffffffu) > 0x7f800000u && (b & 0x7fffffffu) > 0x7f800000u;
    if (a != b && !nan_both) {
      ++mism;
      atomicMin(first, (unsigned)i);
    }
  }
  if (mism) atomicAdd(bad, mism);
}

int main() {
  unsigned long long* bad;
  unsigned* first;
  hipMalloc(&bad, 8);
  hipMalloc(&first, 4);
  hipMemset(bad, 0, 8);
  hipMemset(first, 0xff, 4);
  hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, bad, first);
  unsigned long long h = 0;
  unsigned f = 0;
  hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost);
  hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
  printf("mismatches (NaN payloads aside): %llu of 4294967296%s", h, h ? "" : "\n");
  if (h) printf("; first at bits 0x%08x\n", f);
  return h != 0;
}
