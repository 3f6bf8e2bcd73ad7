// Where does the large-shape NT plane GEMM (csrc/pgemm_big.hip) spend its time?  One K chunk of configs[4]'s input projection
// (3072 x 36 864 x 4096, three passes, both operands as images) with parts of the kernel switched off at compile time:
//   for a in 0 1 2 3 4 5 7 8; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -DBG_ABLATE=$a \
//       tools/gemm256_ablate.hip -o /tmp/g256_$a && /tmp/g256_$a; done
// BG_ABLATE bits: 1 no steady-state LDS-DMA, 2 no fragment reads after the first, 4 no MFMAs, 8 no barriers (results are
// garbage in every ablated build: timing only).  The operands are random fp16 (DVFS: zeros run faster).
#include "../windgnn_amd/csrc/pgemm_big.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
bool g_prof_on = false;
void prof_begin(const char*, double, double, hipStream_t) {}
void prof_end(hipStream_t) {}
int opt_big_gemm() { return 1; }

int main() {
  const int M = 3072, N = 36864, K = 4096, Np = N + 416;
  const size_t na = (size_t)2 * M * K, nb = (size_t)2 * Np * K;
  std::vector<_Float16> h(1 << 22);
  srand(1);
  for (auto& x : h) x = (_Float16)((rand() & 0xffff) / 65536.f - 0.5f);
  _Float16 *A, *B;
  float* C;
  (void)hipMalloc(&A, na * 2);
  (void)hipMalloc(&B, nb * 2);
  (void)hipMalloc(&C, (size_t)M * N * 4);
  for (size_t o = 0; o < na; o += h.size()) (void)hipMemcpy(A + o, h.data(), (na - o < h.size() ? na - o : h.size()) * 2, hipMemcpyHostToDevice);
  for (size_t o = 0; o < nb; o += h.size()) (void)hipMemcpy(B + o, h.data(), (nb - o < h.size() ? nb - o : h.size()) * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipEventRecord(e0);
    const int rc = launch_pgemm_nt256(A, A + (size_t)M * K, M, 0, K, B, Np, (size_t)Np * K, C, N, N, rep > 0, 0);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rc != 0) { printf("launch failed %d\n", rc); return 1; }
    if (rep > 1 && ms < best) best = ms;
  }
  const double fl = 2.0 * M * (double)N * K * 3;
  printf("BG_ABLATE=%d%s%s%s%s: %.3f ms per K chunk  = %.2f PFLOP/s issued (x13 chunks = %.1f ms for GI)\n", BG_ABLATE,
         (BG_ABLATE & 1) ? " noDMA" : "", (BG_ABLATE & 2) ? " noLDSread" : "", (BG_ABLATE & 4) ? " noMFMA" : "",
         (BG_ABLATE & 8) ? " noBarrier" : "", best, fl / best / 1e12, best * 13);
  return 0;
}
