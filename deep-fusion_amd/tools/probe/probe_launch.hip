// probe_launch.hip -- per-launch overhead of back-to-back launches on one stream for the launch
// shapes of the conv kernels (persistent grid of 256 workgroups): empty kernels, HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty(int *p) { extern __shared__ int sm[]; if (p && threadIdx.x == 4095) p[0] = sm[0]; }
int main() {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int shapes[][3] = {{256, 1024, 116 * 1024}, {256, 1024, 0}, {256, 256, 0}, {512, 256, 64 * 1024}, {1, 64, 0}};
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_empty), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (auto &s : shapes) {
    for (int i = 0; i < 20; ++i) k_empty<<<s[0], s[1], s[2]>>>(nullptr);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int it = 500;
    for (int i = 0; i < it; ++i) k_empty<<<s[0], s[1], s[2]>>>(nullptr);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("grid %4d block %4d lds %6d: %.2f us per launch\n", s[0], s[1], s[2], ms * 1e3 / it);
  }
  return 0;
}
