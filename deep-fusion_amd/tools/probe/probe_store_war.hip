// probe_store_war.hip -- may a VALU instruction overwrite the data registers of a global store issued
// just before it?  (hipcc: 1 wait state for stores wider than 64 bits, none otherwise.)  Every wave
// streams stores to its own region (the chip is store-bound, so the memory pipe is backed up) in
// the shape of the conv kernel's epilogue: data built in registers, stored, registers immediately
// rewritten with the NEXT value.  Afterwards the buffer is checked.  K = wait states between the
// store and the first overwrite.  build: hipcc -O2 --offload-arch=gfx950 probe_store_war.hip -o probe_store_war
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(2); } } while (0)

// value stored at 16-byte slot i of a wave's region: word j = tag(i) + j
__device__ __host__ inline int tagv(long long slot, int j) { return (int)(slot * 2654435761u) + j * 0x01010101; }

#define WAR_KERNEL(NAME, STORE, NW, GAP)                                                          \
  __global__ __launch_bounds__(1024) void NAME(int *buf, int per_wave_iters) {                    \
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;               \
    const int lane = threadIdx.x & 63;                                                            \
    int *p = buf + (wave * per_wave_iters * 64 + lane) * NW;                                      \
    int x0, x1, x2, x3, n0, n1, n2, n3;                                                           \
    long long slot = wave * per_wave_iters * 64 + lane;                                           \
    x0 = tagv(slot, 0); x1 = tagv(slot, 1); x2 = tagv(slot, 2); x3 = tagv(slot, 3);               \
    for (int it = 0; it < per_wave_iters; ++it) {                                                 \
      const long long ns = slot + 64;                                                             \
      n0 = tagv(ns, 0); n1 = tagv(ns, 1); n2 = tagv(ns, 2); n3 = tagv(ns, 3);                     \
      asm volatile("v_mov_b32 v100, %1\n\tv_mov_b32 v101, %2\n\tv_mov_b32 v102, %3\n\tv_mov_b32 v103, %4\n\t" \
                   "s_nop 1\n\t" STORE GAP                                                        \
                   "v_mov_b32 v103, %8\n\tv_mov_b32 v101, %6\n\tv_mov_b32 v102, %7\n\tv_mov_b32 v100, %5\n\t" \
                   :                                                                              \
                   : "v"(p), "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(n0), "v"(n1), "v"(n2), "v"(n3) \
                   : "memory", "v100", "v101", "v102", "v103");                                   \
      x0 = n0; x1 = n1; x2 = n2; x3 = n3;                                                         \
      slot = ns;                                                                                  \
      p += 64 * NW;                                                                               \
    }                                                                                             \
  }
WAR_KERNEL(k_x4_0, "global_store_dwordx4 %0, v[100:103], off nt\n\t", 4, "")
WAR_KERNEL(k_x4_1, "global_store_dwordx4 %0, v[100:103], off nt\n\t", 4, "s_nop 0\n\t")
WAR_KERNEL(k_x4_2, "global_store_dwordx4 %0, v[100:103], off nt\n\t", 4, "s_nop 1\n\t")
WAR_KERNEL(k_x4_4, "global_store_dwordx4 %0, v[100:103], off nt\n\t", 4, "s_nop 3\n\t")
WAR_KERNEL(k_p4_0, "global_store_dwordx4 %0, v[100:103], off\n\t", 4, "")
WAR_KERNEL(k_p4_1, "global_store_dwordx4 %0, v[100:103], off\n\t", 4, "s_nop 0\n\t")
WAR_KERNEL(k_p4_2, "global_store_dwordx4 %0, v[100:103], off\n\t", 4, "s_nop 1\n\t")
WAR_KERNEL(k_x3_1, "global_store_dwordx3 %0, v[100:102], off nt\n\t", 4, "s_nop 0\n\t")
WAR_KERNEL(k_x2_0, "global_store_dwordx2 %0, v[102:103], off nt\n\t", 2, "")
WAR_KERNEL(k_x2_1, "global_store_dwordx2 %0, v[102:103], off nt\n\t", 2, "s_nop 0\n\t")
WAR_KERNEL(k_x1_0, "global_store_dword %0, v103, off nt\n\t", 1, "")
WAR_KERNEL(k_x1_1, "global_store_dword %0, v103, off nt\n\t", 1, "s_nop 0\n\t")

// the same for LDS: ds_write_b128 followed by an overwrite of its data registers; the wave reads its
// own slot back and compares (other waves' LDS traffic backs the LDS pipe up)
#define LDS_KERNEL(NAME, GAP)                                                                     \
  __global__ __launch_bounds__(1024) void NAME(int *buf, int per_wave_iters) {                    \
    __shared__ __attribute__((aligned(16))) int lds[16 * 64 * 4 * 8];                             \
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;                                   \
    unsigned long long bad = 0;                                                                   \
    for (int it = 0; it < per_wave_iters * 16; ++it) {                                            \
      const int slot = it & 7;                                                                    \
      const unsigned addr = (unsigned)(size_t)(lds + ((wave * 8 + slot) * 64 + lane) * 4) & 0x3ffffu; \
      const int x0 = it * 4 + lane, x1 = x0 + 1000, x2 = x0 + 2000, x3 = x0 + 3000;               \
      int r0, r1, r2, r3;                                                                         \
      asm volatile("v_mov_b32 v100, %5\n\tv_mov_b32 v101, %6\n\tv_mov_b32 v102, %7\n\tv_mov_b32 v103, %8\n\t" \
                   "s_nop 1\n\tds_write_b128 %4, v[100:103]\n\t" GAP                              \
                   "v_mov_b32 v103, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v100, 0\n\t" \
                   "ds_read_b128 v[104:107], %4\n\ts_waitcnt lgkmcnt(0)\n\t"                      \
                   "v_mov_b32 %0, v104\n\tv_mov_b32 %1, v105\n\tv_mov_b32 %2, v106\n\tv_mov_b32 %3, v107\n\t" \
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)                                   \
                   : "v"(addr), "v"(x0), "v"(x1), "v"(x2), "v"(x3)                                \
                   : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107");   \
      if (r0 != x0 || r1 != x1 || r2 != x2 || r3 != x3) ++bad;                                    \
    }                                                                                             \
    if (bad) atomicAdd((unsigned long long *)buf, bad);                                           \
  }
LDS_KERNEL(k_l4_0, "")
LDS_KERNEL(k_l4_1, "s_nop 0\n\t")
LDS_KERNEL(k_l4_2, "s_nop 1\n\t")
LDS_KERNEL(k_l4_3, "s_nop 2\n\t")
LDS_KERNEL(k_l4_4, "s_nop 3\n\t")
LDS_KERNEL(k_l4_6, "s_nop 5\n\t")
LDS_KERNEL(k_l4_8, "s_nop 7\n\t")
LDS_KERNEL(k_l4_12, "s_nop 11\n\t")
LDS_KERNEL(k_l4_16, "s_nop 15\n\t")

// two stores back to back (the conv epilogue stores pixel pairs): A = v[100:103] -> slot, B = v[104:107]
// -> slot + 64 (next iteration's slot is + 128); then GAP; then both data sets AND the address
// registers are overwritten.  MID = what sits between the two stores.
#define WAR2_KERNEL(NAME, MID, GAP)                                                               \
  __global__ __launch_bounds__(1024) void NAME(int *buf, int per_wave_iters) {                    \
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;               \
    const int lane = threadIdx.x & 63;                                                            \
    int *p = buf + (wave * per_wave_iters * 64 + lane) * 4;                                       \
    long long slot = wave * per_wave_iters * 64 + lane;                                           \
    for (int it = 0; it < per_wave_iters; it += 2) {                                              \
      int *q = p + 64 * 4;                                                                        \
      const int a0 = tagv(slot, 0), a1 = tagv(slot, 1), a2 = tagv(slot, 2), a3 = tagv(slot, 3);   \
      const int b0 = tagv(slot + 64, 0), b1 = tagv(slot + 64, 1), b2 = tagv(slot + 64, 2), b3 = tagv(slot + 64, 3); \
      asm volatile("v_mov_b32 v100, %2\n\tv_mov_b32 v101, %3\n\tv_mov_b32 v102, %4\n\tv_mov_b32 v103, %5\n\t" \
                   "v_mov_b32 v104, %6\n\tv_mov_b32 v105, %7\n\tv_mov_b32 v106, %8\n\tv_mov_b32 v107, %9\n\t" \
                   "v_mov_b32 v108, %0\n\tv_mov_b32 v109, %1\n\tv_mov_b32 v110, %10\n\tv_mov_b32 v111, %11\n\t" \
                   "s_nop 1\n\t"                                                                  \
                   "global_store_dwordx4 v[108:109], v[100:103], off nt\n\t" MID                   \
                   "global_store_dwordx4 v[110:111], v[104:107], off nt\n\t" GAP                   \
                   "v_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\t" \
                   "v_mov_b32 v101, 0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v100, 0\n\tv_mov_b32 v102, 0\n\t" \
                   "v_mov_b32 v105, 0\n\tv_mov_b32 v107, 0\n\tv_mov_b32 v104, 0\n\tv_mov_b32 v106, 0\n\t" \
                   :                                                                              \
                   : "v"((unsigned)(size_t)p), "v"((unsigned)((size_t)p >> 32)), "v"(a0), "v"(a1), "v"(a2), "v"(a3), \
                     "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"((unsigned)(size_t)q), "v"((unsigned)((size_t)q >> 32)) \
                   : "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111"); \
      slot += 128;                                                                                \
      p += 128 * 4;                                                                               \
    }                                                                                             \
  }
WAR2_KERNEL(k_2_a, "", "")
WAR2_KERNEL(k_2_b, "", "s_nop 0\n\t")
WAR2_KERNEL(k_2_c, "", "s_nop 1\n\t")
WAR2_KERNEL(k_2_d, "s_nop 1\n\t", "s_nop 1\n\t")
WAR2_KERNEL(k_2_e, "", "s_nop 3\n\t")

typedef void (*kern_t)(int *, int);
struct Cfg { const char *name; kern_t k; int nw, states; };
int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int grid = prop.multiProcessorCount, waves = grid * 16, iters = 512;
  const size_t words = (size_t)waves * iters * 64 * 4;
  int *buf;
  CK(hipMalloc(&buf, words * 4));
  std::vector<int> h(words);
  const Cfg cfgs[] = {{"global_store_dwordx4", k_x4_0, 4, 0}, {"global_store_dwordx4", k_x4_1, 4, 1}, {"global_store_dwordx4", k_x4_2, 4, 2},
                      {"global_store_dwordx4", k_x4_4, 4, 4}, {"global_store_dwordx4 (no nt)", k_p4_0, 4, 0}, {"global_store_dwordx4 (no nt)", k_p4_1, 4, 1}, {"global_store_dwordx4 (no nt)", k_p4_2, 4, 2},
                      {"global_store_dwordx2", k_x2_0, 2, 0}, {"global_store_dwordx2", k_x2_1, 2, 1},
                      {"global_store_dword", k_x1_0, 1, 0}, {"global_store_dword", k_x1_1, 1, 1}};
  for (const Cfg &c : cfgs) {
    CK(hipMemset(buf, 0xEE, words * 4));
    for (int rep = 0; rep < 3; ++rep) c.k<<<grid, 1024>>>(buf, iters);
    CK(hipDeviceSynchronize());
    const size_t n = (size_t)waves * iters * 64 * c.nw;
    CK(hipMemcpy(h.data(), buf, n * 4, hipMemcpyDeviceToHost));
    unsigned long long bad = 0;
    for (size_t i = 0; i < n; ++i) {
      const long long slot = (long long)(i / c.nw);
      const int j = c.nw == 4 ? (int)(i % 4) : c.nw == 2 ? 2 + (int)(i % 2) : 3;
      if (h[i] != tagv(slot, j)) ++bad;
    }
    printf("{\"probe\": \"store_data_war\", \"store\": \"%s\", \"wait_states_before_overwrite\": %d, \"stores\": %.3g, \"wrong_words\": %llu, \"MB\": %.0f}\n",
           c.name, c.states, (double)waves * iters * 64 * 3, bad, n * 4 / 1e6);
    fflush(stdout);
  }
  {
    const kern_t k2[] = {k_2_a, k_2_b, k_2_c, k_2_d, k_2_e};
    const char *n2[] = {"A; B; overwrite addresses+data at once", "A; B; s_nop 0; overwrite", "A; B; s_nop 1; overwrite",
                        "A; s_nop 1; B; s_nop 1; overwrite (what dfx_store16 emits)", "A; B; s_nop 3; overwrite"};
    for (int m = 0; m < 5; ++m) {
      CK(hipMemset(buf, 0xEE, words * 4));
      for (int rep = 0; rep < 3; ++rep) k2[m]<<<grid, 1024>>>(buf, iters);
      CK(hipDeviceSynchronize());
      const size_t n = (size_t)waves * iters * 64 * 4;
      CK(hipMemcpy(h.data(), buf, n * 4, hipMemcpyDeviceToHost));
      unsigned long long bad = 0;
      for (size_t i = 0; i < n; ++i)
        if (h[i] != tagv((long long)(i / 4), (int)(i % 4))) ++bad;
      printf("{\"probe\": \"store_data_war\", \"store\": \"two global_store_dwordx4: %s\", \"stores\": %.3g, \"wrong_words\": %llu}\n",
             n2[m], (double)waves * iters * 64 * 3, bad);
      fflush(stdout);
    }
  }
  const kern_t lk[] = {k_l4_0, k_l4_1, k_l4_2, k_l4_3, k_l4_4, k_l4_6, k_l4_8, k_l4_12, k_l4_16};
  const int lstates[] = {0, 1, 2, 3, 4, 6, 8, 12, 16};
  for (int mi = 0; mi < 9; ++mi) {
    const int m = lstates[mi];
    CK(hipMemset(buf, 0, 8));
    lk[mi]<<<grid, 1024>>>(buf, iters);
    CK(hipDeviceSynchronize());
    unsigned long long b;
    CK(hipMemcpy(&b, buf, 8, hipMemcpyDeviceToHost));
    printf("{\"probe\": \"store_data_war\", \"store\": \"ds_write_b128\", \"wait_states_before_overwrite\": %d, \"stores\": %.3g, \"wrong_lanes\": %llu}\n",
           m, (double)waves * iters * 16 * 64, b);
  }
  return 0;
}
