#include <hip/hip_runtime.h>
__device__ __forceinline__ float shl1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); }
__device__ __forceinline__ float shr1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); }
__global__ void k(const float* a, float* o) {
  float t0 = a[threadIdx.x], t2 = a[64 + threadIdx.x], t4 = a[128 + threadIdx.x];
  float l = shl1(shl1(t0) + t2) + t4 + shr1(t2);
  o[threadIdx.x] = l;
}
int main() {
  float *a, *o; hipMalloc(&a, 192 * 4); hipMalloc(&o, 256);
  float h[192]; for (int i = 0; i < 192; ++i) h[i] = i;
  hipMemcpy(a, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(a, o);
  float r[64]; hipMemcpy(r, o, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) {
    float t0 = i + 2 < 64 ? h[i + 2] : 0, t2 = i + 1 < 64 ? h[64 + i + 1] : 0, t4 = h[128 + i], s = i >= 1 ? h[64 + i - 1] : 0;
    // shl1(shl1(t0)+t2): inner value at lane i+1 is (t0[i+2] (0 if i+2>63)) + t2[i+1]; whole is 0 if i+1 > 63
    float want = (i + 1 < 64 ? t0 + t2 : 0) + t4 + s;
    if (r[i] != want) { ++bad; printf("lane %d got %f want %f\n", i, r[i], want); }
  }
  printf("bad=%d\n", bad);
}
