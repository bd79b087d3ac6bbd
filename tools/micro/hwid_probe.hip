// Where do the workgroups / wavefronts of a fully resident grid land?  Dumps (XCC_ID, HW_ID) per wavefront of a 1024 x 256-thread
// grid limited to 4 workgroups per CU by LDS (the residency of k_sattn_*).   hipcc --offload-arch=gfx950 -O2 hwid_probe.hip -o hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void k(unsigned* out) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if ((threadIdx.x & 63) == 0) {
    const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
    out[2 * w] = hw; out[2 * w + 1] = xcc;
  }
  lds[threadIdx.x] = 0.f;
  // stay resident for a while so that the whole grid co-resides
  for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);
}
int main() {
  const int B = 1024;
  unsigned* d; hipMalloc(&d, B * 4 * 2 * sizeof(unsigned));
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024);
  k<<<B, 256, 40 * 1024>>>(d);
  std::vector<unsigned> h(B * 8);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  // decode: gfx9 HW_ID: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]
  std::map<unsigned, std::vector<int>> cu_blocks;
  for (int w = 0; w < B * 4; ++w) {
    const unsigned hw = h[2 * w], xcc = h[2 * w + 1] & 15;
    const unsigned simd = (hw >> 4) & 3, cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    if ((w & 3) == 0) cu_blocks[key].push_back(w / 4);
    if (w < 64) printf("wave %4d block %3d w%d -> xcc %u se %u sh %u cu %2u simd %u slot %u\n", w, w / 4, w & 3, xcc, se, sh, cu, simd, hw & 15);
  }
  printf("distinct CUs: %zu\n", cu_blocks.size());
  int shown = 0;
  for (auto& kv : cu_blocks) {
    if (shown++ < 24) { printf("cu %05x blocks:", kv.first); for (int b : kv.second) printf(" %d", b); printf("\n"); }
  }
  // histogram of blocks per CU
  std::map<size_t, int> hist; for (auto& kv : cu_blocks) hist[kv.second.size()]++;
  for (auto& kv : hist) printf("%d CUs hold %zu blocks\n", kv.second, kv.first);
  return 0;
}
