// Micro-benchmark for VERDICT r4 weak 4 / next 2(a): the B stream of the fused front end's GEMM waves (csrc/gcngi.hip), alone.
//
// In gcngi_fwd_kernel every CU re-streams the SAME 573 KB image of W_ih (hi + lo planes, stage-major: [K/32][Np = 320][32]
// halfs, so one 16-column fragment of one K step is 1 KB contiguous) from L2 into registers once per 32-row tile: measured
// 23 B/clk/CU, against the 54-58 B/clk/CU tools/l2_rate.hip reads out of an L2-resident window with the same 1 KB
// wave-instructions.  This tool reproduces the GEMM waves' access stream and nothing else and varies what could separate the
// two numbers:
//   W     waves per CU that stream (each owns CT column tiles of the 20; tiles wrap around when W * CT > 20)
//   CT    column tiles per wave and pass;  PL planes (2 = hi + lo)      -> CT * PL loads of 1 KB per wave and K step
//   D     K steps requested ahead (register stages in flight = D; the shipped kernel: 1)
//   MF    MFMAs (16x16x32 f16) issued per K step on the loaded fragments (0 = loads only; the shipped f16x3 kernel: RT * CT * 3 = 18)
//   PAT   0: every CU walks the same image in step (the shipped kernel)
//         1: every CU starts at its own K step (blockIdx * 5 % nk) and wave (staggered: same bytes, different moments)
//         2: every CU has a private copy of the image (256 x 573 KB = 147 MB: served by the Infinity Cache, not L2)
//         3: the control of tools/l2_rate.hip -- 64 KB blocks hopping through a 2 MB window -- with this kernel's loop
// Output: us per launch, B/clk/CU at the clock measured in the kernel (s_memtime / s_memrealtime), bytes in flight per CU
// that the setting implies (W * D * CT * PL KB).
//   hipcc --offload-arch=gfx950 -O3 tools/l2_stream.hip -o /tmp/l2_stream && /tmp/l2_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NK = 14, NP = 320, NCT = 20;                      // K steps, plane rows, column tiles (3H = 306 -> 320)
constexpr size_t KSTEP_B = (size_t)NP * 64;                     // bytes per K step and plane
constexpr size_t PLANE_B = KSTEP_B * NK;                        // 286 720
constexpr size_t IMAGE_B = 2 * PLANE_B;                         // 573 440

template <int W, int CT, int PL, int D, int MF, int PAT>
__global__ void __launch_bounds__(64 * W) stream_kernel(const char* __restrict__ img, int ntiles, float* sink,
                                                         unsigned long long* clocks) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = img + (PAT == 2 ? (size_t)blockIdx.x * IMAGE_B : 0);
  const int r16 = lane & 15, c4 = lane >> 4;
  unsigned loff[CT];
#pragma unroll
  for (int j = 0; j < CT; ++j) loff[j] = (unsigned)(((((wave * CT + j) % NCT) * 16 + r16) * 32 + 8 * c4) * 2);
  const int k0 = PAT == 1 ? (int)((blockIdx.x * 5 + wave * 3) % NK) : 0;
  f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const h8 a = {(_Float16)1.f, (_Float16)0.f, (_Float16)1.f, (_Float16)0.f, (_Float16)1.f, (_Float16)0.f, (_Float16)1.f, (_Float16)0.f};
  h8 st[D + 1][CT * PL];
  const int total = ntiles * NK;
  auto issue = [&](h8 (&s)[CT * PL], int idx) {
    if (idx >= total) idx = total - 1;
    const char* p;
    if (PAT == 3) {           // l2_rate's walk: 64 KB blocks of a 2 MB window, 1 KB per wave-instruction
      const size_t blk = ((size_t)blockIdx.x * 7 + (size_t)idx * 13) % 32;
      p = base + blk * 65536 + (size_t)((wave * CT * PL) % 64) * 1024 + lane * 16;
#pragma unroll
      for (int j = 0; j < CT * PL; ++j) s[j] = *(const h8*)(p + (size_t)(j % 8) * 1024);
    } else {
      const int kt = (idx + k0) % NK;
      p = base + (size_t)kt * KSTEP_B;
#pragma unroll
      for (int j = 0; j < CT; ++j) {
        s[j * PL] = *(const h8*)(p + loff[j]);
        if (PL == 2) s[j * PL + 1] = *(const h8*)(p + PLANE_B + loff[j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto consume = [&](const h8 (&s)[CT * PL]) {
    if (MF == 0) {
#pragma unroll
      for (int j = 0; j < CT * PL; ++j) acc[j & 3] += __builtin_convertvector(__builtin_shufflevector(s[j], s[j], 0, 1, 2, 3), f32x4);
    } else {
#pragma unroll
      for (int m = 0; m < MF; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, s[m % (CT * PL)], acc[m & 3], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
  for (int d = 0; d < D; ++d) issue(st[d], d);
  for (int idx = 0; idx < total; idx += D + 1) {
#pragma unroll
    for (int s = 0; s <= D; ++s) {
      issue(st[(s + D) % (D + 1)], idx + s + D);
      consume(st[s]);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    clocks[2 * blockIdx.x] = t1 - t0;
    clocks[2 * blockIdx.x + 1] = r1 - r0;
  }
  const f32x4 v = acc[0] + acc[1] + acc[2] + acc[3];
  if (v[0] + v[1] + v[2] + v[3] == 12345.678f) sink[threadIdx.x] = v[0];
}

static char* g_img;
static float* g_sink;
static unsigned long long* g_clk;

template <int W, int CT, int PL, int D, int MF, int PAT>
static void run() {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int ntiles = 12;                                          // 12 tiles per CU = B * T = 98 304 rows / 32 / 256
  // total K steps must be a multiple of D + 1 for the unrolled ring: 168 = 12 * 14 is divisible by 1..4 and 6, 7, 8
  float ms = 0.f, best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(a);
    hipLaunchKernelGGL((stream_kernel<W, CT, PL, D, MF, PAT>), dim3(256), dim3(64 * W), 0, 0, g_img, ntiles, g_sink, g_clk);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  unsigned long long h[512];
  (void)hipMemcpy(h, g_clk, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0, real = 0;
  for (int i = 0; i < 256; ++i) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
  const double ghz = cyc / real * 0.1;                             // s_memrealtime ticks at 100 MHz
  const double bytes_cu = (double)ntiles * NK * W * CT * PL * 1024.0;
  static const char* pat[] = {"same image, in step", "same image, staggered", "private copies (MALL)", "2 MB window, 64 KB hops"};
  printf("W=%2d CT=%d PL=%d D=%d MF=%2d  %-24s in flight %3d KB/CU  %7.1f us  %5.1f B/clk/CU (in-kernel %.2f GHz, %5.1f B/clk by cycles)  %5.2f TB/s\n",
         W, CT, PL, D, MF, pat[PAT], W * D * CT * PL, best * 1e3, bytes_cu / (best * 1e-3 * ghz * 1e9), ghz,
         bytes_cu / (cyc / 256.0), bytes_cu * 256 / best / 1e9);
  fflush(stdout);
}

int main() {
  (void)hipMalloc(&g_img, (size_t)256 * IMAGE_B + (2 << 20));
  (void)hipMalloc(&g_sink, 1 << 16);
  (void)hipMalloc(&g_clk, 512 * 8);
  (void)hipMemset(g_img, 0x11, (size_t)256 * IMAGE_B + (2 << 20));
  printf("# B stream of gcngi's GEMM waves, alone: 12 tiles x 14 K steps per CU, 1 KB per wave-load, 256 CUs\n");
  printf("# --- the shipped setting (8 waves x 3 tiles x 2 planes, one K step ahead), loads only / with its 18 MFMAs per K step\n");
  run<8, 3, 2, 1, 0, 0>(); run<8, 3, 2, 1, 18, 0>();
  printf("# --- controls: who reads what when (loads only, then with MFMAs)\n");
  run<8, 3, 2, 1, 0, 1>(); run<8, 3, 2, 1, 0, 2>(); run<8, 3, 2, 1, 0, 3>();
  run<8, 3, 2, 1, 18, 1>(); run<8, 3, 2, 1, 18, 2>(); run<8, 3, 2, 1, 18, 3>();
  printf("# --- K steps requested ahead (bytes in flight), loads only\n");
  run<8, 3, 2, 2, 0, 0>(); run<8, 3, 2, 3, 0, 0>(); run<8, 3, 2, 5, 0, 0>();
  printf("# --- the same with the MFMAs\n");
  run<8, 3, 2, 2, 18, 0>(); run<8, 3, 2, 3, 18, 0>();
  printf("# --- waves per CU (loads only; CT = 2: 12 waves x 2 tiles cover 24 >= 20 tiles)\n");
  run<4, 3, 2, 1, 0, 0>(); run<12, 2, 2, 1, 0, 0>(); run<12, 2, 2, 2, 0, 0>(); run<16, 2, 2, 1, 0, 0>(); run<16, 2, 2, 2, 0, 0>();
  run<16, 3, 2, 1, 0, 0>();
  printf("# --- column tiles per pass (loads only, then MFMAs scaled: RT * CT * 3)\n");
  run<8, 1, 2, 1, 0, 0>(); run<8, 2, 2, 1, 0, 0>(); run<8, 1, 2, 3, 0, 0>(); run<8, 2, 2, 2, 0, 0>();
  run<8, 1, 2, 1, 6, 0>(); run<8, 2, 2, 1, 12, 0>(); run<8, 2, 2, 2, 12, 0>(); run<8, 2, 2, 3, 12, 0>();
  printf("# --- one-pass mode (one plane; shipped: 4 waves x 5 tiles in passes of 3 + 2, 48-row tiles: MF = 3 * CT)\n");
  run<4, 3, 1, 1, 0, 0>(); run<4, 3, 1, 1, 9, 0>(); run<4, 3, 1, 2, 9, 0>(); run<4, 3, 1, 3, 9, 0>(); run<8, 3, 1, 1, 9, 0>();
  run<8, 3, 1, 2, 9, 0>();
  printf("# --- 16 waves with MFMAs (a phase-split kernel: every wave a GEMM wave)\n");
  run<16, 2, 2, 1, 12, 0>(); run<16, 2, 2, 2, 12, 0>(); run<16, 1, 2, 3, 6, 0>();
  return 0;
}
