// Dev tool: where do the workgroups of a kernel land when its stream carries a CU mask
// (hipExtStreamCreateWithCUMask)?  Prints, per mask, the number of workgroups seen per XCC and the
// distinct (XCC, SE, CU) triples used.  Build: hipcc --offload-arch=gfx950 -O2 cu_mask_census.hip -o cu_mask_census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void census(uint32_t* out, int spin) {
  uint32_t xcc, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  // stay resident for a while so that the grid spreads over every CU the mask allows
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
}

static int run(const char* name, hipStream_t st, uint32_t* dbuf, int nwg, size_t lds) {
  CK(hipMemsetAsync(dbuf, 0xff, nwg * 8, st));
  hipLaunchKernelGGL(census, dim3(nwg), dim3(256), lds, st, dbuf, 2000);   // 20 us resident
  CK(hipStreamSynchronize(st));
  std::vector<uint32_t> h(2 * nwg);
  CK(hipMemcpy(h.data(), dbuf, nwg * 8, hipMemcpyDeviceToHost));
  std::map<int, int> per_xcc; std::set<uint32_t> cus; std::map<int, std::set<uint32_t>> cus_per_xcc;
  for (int i = 0; i < nwg; ++i) {
    const int xcc = h[2 * i] & 0xf; const uint32_t hw = h[2 * i + 1];
    const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    per_xcc[xcc]++; cus.insert(xcc << 16 | se << 8 | sh << 4 | cu); cus_per_xcc[xcc].insert(se << 8 | sh << 4 | cu);
  }
  printf("%-28s %d WGs on %zu distinct CUs; per XCC (WGs/CUs):", name, nwg, cus.size());
  for (auto& kv : per_xcc) printf(" %d:%d/%zu", kv.first, kv.second, cus_per_xcc[kv.first].size());
  printf("\n   first 16 WGs -> xcc: ");
  for (int i = 0; i < 16; ++i) printf("%d ", h[2 * i] & 0xf);
  printf("\n");
  return 0;
}

int main() {
  uint32_t* dbuf; CK(hipMalloc(&dbuf, 4096 * 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&census), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  hipStream_t plain; CK(hipStreamCreate(&plain));
  run("no mask, 160 KB LDS", plain, dbuf, 256, 160 * 1024);
  run("no mask, 128 WGs 160 KB", plain, dbuf, 128, 160 * 1024);
  run("no mask, 256 WGs 58 KB", plain, dbuf, 256, 58 * 1024);
  struct M { const char* name; uint32_t m[8]; } masks[] = {
    {"low 128 bits", {~0u, ~0u, ~0u, ~0u, 0, 0, 0, 0}},
    {"high 128 bits", {0, 0, 0, 0, ~0u, ~0u, ~0u, ~0u}},
    {"bits b%8<4", {0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu}},
    {"bits b%8>=4", {0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u, 0xf0f0f0f0u}},
    {"even bits", {0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u, 0x55555555u}},
    {"low 32 bits", {~0u, 0, 0, 0, 0, 0, 0, 0}},
  };
  for (auto& m : masks) {
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, m.m));
    char nm[64]; snprintf(nm, sizeof nm, "%s, 160 KB", m.name);
    if (run(nm, s, dbuf, 128, 160 * 1024)) return 1;
    snprintf(nm, sizeof nm, "%s, 58 KB x256", m.name);
    if (run(nm, s, dbuf, 256, 58 * 1024)) return 1;
    CK(hipStreamDestroy(s));
  }
  return 0;
}
