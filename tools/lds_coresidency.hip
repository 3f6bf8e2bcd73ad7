// How many workgroups of a given LDS size / VGPR count does one CU of gfx950 hold at the same time?  (Round 5: the exact-fp32 NT
// GEMM as two 4-wave workgroups of 72 KB LDS each was no faster than one 8-wave workgroup of 114 KB -- were they co-resident?)
// Every workgroup records the CU it ran on (XCC_ID, HW_ID) and its start / end time (s_memrealtime, 100 MHz); the host counts,
// per CU, the largest number of workgroups whose intervals overlap.
//   hipcc --offload-arch=gfx950 -O3 tools/lds_coresidency.hip -o /tmp/lds_cores && /tmp/lds_cores
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

template <int VG>
__global__ void __launch_bounds__(256) probe(unsigned long long* rec, int spin_ticks, float* sink) {
  extern __shared__ char smem[];
  float keep[VG];
#pragma unroll
  for (int i = 0; i < VG; ++i) keep[i] = (float)(threadIdx.x + i);
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  smem[threadIdx.x] = 1;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) {
#pragma unroll
    for (int i = 0; i < VG; ++i) keep[i] = keep[i] * 1.0001f + 0.5f;
    __builtin_amdgcn_s_sleep(8);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < VG; ++i) s += keep[i];
  if (s == 1234.5f) sink[0] = s + smem[threadIdx.x];
  if (threadIdx.x == 0) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    rec[3 * blockIdx.x] = ((unsigned long long)(xcc & 15) << 12) | (se << 8) | (sh << 4) | cu;
    rec[3 * blockIdx.x + 1] = t0;
    rec[3 * blockIdx.x + 2] = t1;
  }
}

template <int VG>
static void run(int lds, int nwg) {
  unsigned long long* d;
  float* sink;
  (void)hipMalloc(&d, nwg * 24);
  (void)hipMalloc(&sink, 64);
  (void)hipFuncSetAttribute((const void*)probe<VG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  int occ = -1;
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)probe<VG>, 256, lds);
  hipLaunchKernelGGL(probe<VG>, dim3(nwg), dim3(256), lds, 0, d, 2000 /* 20 us */, sink);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed at lds=%d\n", lds); return; }
  std::vector<unsigned long long> h(3 * nwg);
  (void)hipMemcpy(h.data(), d, nwg * 24, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
  for (int i = 0; i < nwg; ++i) {
    ev[h[3 * i]].push_back({h[3 * i + 1], +1});
    ev[h[3 * i]].push_back({h[3 * i + 2], -1});
  }
  int worst = 0, best = 1 << 30;
  for (auto& kv : ev) {
    std::sort(kv.second.begin(), kv.second.end());
    int cur = 0, mx = 0;
    for (auto& e : kv.second) { cur += e.second; mx = std::max(mx, cur); }
    worst = std::max(worst, mx);
    best = std::min(best, mx);
  }
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int i = 0; i < nwg; ++i) { tmin = std::min(tmin, h[3 * i + 1]); tmax = std::max(tmax, h[3 * i + 2]); }
  printf("~%3d VGPRs  LDS %6d B  %4d workgroups of 256 threads: %3zu CUs seen, concurrent workgroups per CU min %d max %d (runtime's occupancy answer: %d), whole launch %.1f us\n",
         VG + 4, lds, nwg, ev.size(), best, worst, occ, (tmax - tmin) * 0.01);
  (void)hipFree(d);
  (void)hipFree(sink);
}

int main() {
  for (int lds : {16384, 32768, 57344, 65536, 73728, 81920}) run<16>(lds, 1536);
  for (int lds : {57344, 73728}) run<200>(lds, 1536);
  for (int lds : {57344, 73728}) run<110>(lds, 1536);
  return 0;
}
