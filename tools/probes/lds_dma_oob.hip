// Probe: does buffer_load_dwordx4 ... lds with an out-of-range offset write ZEROS into LDS (or leave it untouched)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const unsigned* src, unsigned* out, unsigned nbytes) {
  __shared__ __attribute__((aligned(16))) unsigned lds[64 * 4 * 2];
  for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) lds[i] = 0xDEADBEEFu;
  __syncthreads();
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, (int)nbytes, 0x00020000);
  // lanes 0..31 in range, lanes 32..63 out of range
  unsigned voff = threadIdx.x < 32 ? threadIdx.x * 16 : 0xFFFFFFF0u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)lds, 16, voff, 0, 0, 0);
  // second instruction: exec-masked lanes (only even lanes active) into the second KiB
  if ((threadIdx.x & 1) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + 256), 16, threadIdx.x * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 4 * 2; i += 64) out[i] = lds[i];
}
int main() {
  std::vector<unsigned> h(64 * 4);
  for (int i = 0; i < 64 * 4; ++i) h[i] = 1000 + i;
  unsigned *d, *o;
  hipMalloc(&d, 64 * 16); hipMalloc(&o, 64 * 4 * 2 * 4);
  hipMemcpy(d, h.data(), 64 * 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 64 * 16);
  std::vector<unsigned> r(64 * 4 * 2);
  hipMemcpy(r.data(), o, r.size() * 4, hipMemcpyDeviceToHost);
  printf("in-range lane 3: %u %u %u %u (expect 1012..1015)\n", r[12], r[13], r[14], r[15]);
  printf("OOB lane 40    : %x %x %x %x (0 = zero-filled, deadbeef = untouched)\n", r[160], r[161], r[162], r[163]);
  printf("masked: lane 2 : %u %u (expect 1008 1009)   lane 3 (inactive): %x %x\n", r[256 + 8], r[256 + 9], r[256 + 12], r[256 + 13]);
  return 0;
}
