// Experiment (round-1 verdict, task 3): the GCN stage as an exact-fp32 packed-VALU kernel -- A in LDS, v_pk_fma_f32
// aggregation, no MFMA, no hi/lo splits.  Forward of both layers for B*T = 98304 tiles of S = 34 stations x 13 features,
// one tile per wavefront, checked against a host reference on a few tiles and timed with events.
//   g = relu(A relu(A (X W1) + b1) W2 ... ) exactly as csrc/gcn32.hip computes it (U = X W first, then A U).
// Lane l = (sg, fp): rows s = 4 sg .. 4 sg + 3 (sg = l / 7 < 9) and the feature pair f' = 2 fp, 2 fp + 1 (fp = l % 7):
// every v_pk_fma_f32 updates one row's pair; X / H rows and A^T rows come from LDS as b128, U rows as b64.
//   hipcc --offload-arch=gfx950 -O3 tools/gcn_valu_proto.hip -o /tmp/gcn_valu && /tmp/gcn_valu
// Result (MI355X): see DESIGN.md section 7.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int S = 34, F = 13, I = S * F, SP = 36, FP = 16, WAVES = 4;

__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ void __launch_bounds__(64 * WAVES) gcn_valu_fwd(int ntiles, const float* __restrict__ A,
                                                          const float* __restrict__ X, const float* __restrict__ W1,
                                                          const float* __restrict__ b1, const float* __restrict__ W2,
                                                          const float* __restrict__ b2, float* __restrict__ G) {
  __shared__ __attribute__((aligned(16))) float sAT[SP * SP];            // A^T[s'][s], zero padded
  __shared__ __attribute__((aligned(16))) float sbuf[WAVES][2][SP * FP];  // per wave: [0] X / H rows, [1] U rows
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < SP * SP; i += 64 * WAVES) {
    const int sp = i / SP, s = i % SP;
    sAT[i] = (sp < S && s < S) ? A[s * S + sp] : 0.f;
  }
  float* xb = sbuf[wave][0];
  float* ub = sbuf[wave][1];
  for (int i = lane; i < 2 * SP * FP; i += 64) xb[i] = 0.f;
  __syncthreads();
  const int sg = lane / 7, fp = lane % 7;
  const bool live = sg < 9;
  const int s0 = live ? 4 * sg : 0, f0 = 2 * fp;
  f32x2 w1[F], w2[F];                                 // W[f][f0], W[f][f0 + 1]
  for (int f = 0; f < F; ++f) {
    w1[f] = f32x2{W1[f * F + f0], f0 + 1 < F ? W1[f * F + f0 + 1] : 0.f};
    w2[f] = f32x2{W2[f * F + f0], f0 + 1 < F ? W2[f * F + f0 + 1] : 0.f};
  }
  const f32x2 bb1 = {b1[f0], f0 + 1 < F ? b1[f0 + 1] : 0.f}, bb2 = {b2[f0], f0 + 1 < F ? b2[f0 + 1] : 0.f};
  const int wave_id = blockIdx.x * WAVES + wave, nwaves = gridDim.x * WAVES;

  for (int tile = wave_id; tile < ntiles; tile += nwaves) {
    const float* x = X + (size_t)tile * I;
    for (int e = lane; e < I; e += 64) xb[(e / F) * FP + e % F] = x[e];   // coalesced load, [s][16] rows in LDS
    wave_fence();
#pragma unroll
    for (int layer = 0; layer < 2; ++layer) {
      const f32x2* w = layer ? w2 : w1;
      // U[s][f' pair] = sum_f in[s][f] W[f][f' pair]
      f32x2 u[4] = {};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const f32x4* row = (const f32x4*)(xb + (s0 + r) * FP);
        const f32x4 q0 = row[0], q1 = row[1], q2 = row[2], q3 = row[3];
        const float in[16] = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3],
                              q2[0], q2[1], q2[2], q2[3], q3[0], q3[1], q3[2], q3[3]};
#pragma unroll
        for (int f = 0; f < F; ++f) u[r] = __builtin_elementwise_fma(f32x2{in[f], in[f]}, w[f], u[r]);
      }
      if (live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) *(f32x2*)(ub + (s0 + r) * FP + f0) = u[r];
      }
      wave_fence();
      // out[s][f' pair] = relu(sum_s' A[s][s'] U[s'][f' pair] + b)
      f32x2 h[4] = {};
#pragma unroll 2
      for (int sp = 0; sp < S; ++sp) {
        const f32x4 a = *(const f32x4*)(sAT + sp * SP + s0);
        const f32x2 uu = *(const f32x2*)(ub + sp * FP + f0);
#pragma unroll
        for (int r = 0; r < 4; ++r) h[r] = __builtin_elementwise_fma(f32x2{a[r], a[r]}, uu, h[r]);
      }
      const f32x2 bb = layer ? bb2 : bb1;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        h[r] += bb;
        h[r][0] = fmaxf(h[r][0], 0.f);
        h[r][1] = fmaxf(h[r][1], 0.f);
      }
      if (layer == 0) {
        if (live) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (s0 + r < S) *(f32x2*)(xb + (s0 + r) * FP + f0) = f32x2{h[r][0], f0 + 1 < F ? h[r][1] : 0.f};
        }
        wave_fence();
      } else if (live) {
        float* g = G + (size_t)tile * I;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (s0 + r < S) {
            g[(s0 + r) * F + f0] = h[r][0];
            if (f0 + 1 < F) g[(s0 + r) * F + f0 + 1] = h[r][1];
          }
      }
    }
    wave_fence();
  }
}

int main() {
  const int ntiles = 4096 * 24;
  std::vector<float> hA(S * S), hX((size_t)ntiles * I), hW1(F * F), hW2(F * F), hb1(F), hb2(F), hG((size_t)ntiles * I);
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX; };
  for (auto& v : hA) v = rnd() / S;
  for (auto& v : hX) v = rnd();
  for (auto& v : hW1) v = rnd() - 0.5f;
  for (auto& v : hW2) v = rnd() - 0.5f;
  for (auto& v : hb1) v = rnd() - 0.5f;
  for (auto& v : hb2) v = rnd() - 0.5f;
  float *dA, *dX, *dW1, *dW2, *db1, *db2, *dG;
  (void)hipMalloc(&dA, hA.size() * 4); (void)hipMalloc(&dX, hX.size() * 4); (void)hipMalloc(&dG, hG.size() * 4);
  (void)hipMalloc(&dW1, F * F * 4); (void)hipMalloc(&dW2, F * F * 4); (void)hipMalloc(&db1, F * 4); (void)hipMalloc(&db2, F * 4);
  (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dW1, hW1.data(), F * F * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dW2, hW2.data(), F * F * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(db1, hb1.data(), F * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(db2, hb2.data(), F * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0.f;
  for (int grid : {512, 1024, 2048}) {
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(gcn_valu_fwd, dim3(grid), dim3(64 * WAVES), 0, 0, ntiles, dA, dX, dW1, db1, dW2, db2, dG);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&ms, e0, e1);
    }
    printf("gcn_valu_fwd grid %d x %d waves: %.1f us for %d tiles\n", grid, WAVES, ms * 1e3, ntiles);
  }
  (void)hipMemcpy(hG.data(), dG, hG.size() * 4, hipMemcpyDeviceToHost);
  double err = 0;
  for (int tile : {0, 1, 777, ntiles - 1}) {
    std::vector<double> U(I), H(I), U2(I);
    const float* x = &hX[(size_t)tile * I];
    for (int s = 0; s < S; ++s) for (int j = 0; j < F; ++j) { double a = 0; for (int f = 0; f < F; ++f) a += (double)x[s * F + f] * hW1[f * F + j]; U[s * F + j] = a; }
    for (int s = 0; s < S; ++s) for (int j = 0; j < F; ++j) { double a = hb1[j]; for (int t = 0; t < S; ++t) a += (double)hA[s * S + t] * U[t * F + j]; H[s * F + j] = a > 0 ? a : 0; }
    for (int s = 0; s < S; ++s) for (int j = 0; j < F; ++j) { double a = 0; for (int f = 0; f < F; ++f) a += H[s * F + f] * hW2[f * F + j]; U2[s * F + j] = a; }
    for (int s = 0; s < S; ++s) for (int j = 0; j < F; ++j) { double a = hb2[j]; for (int t = 0; t < S; ++t) a += (double)hA[s * S + t] * U2[t * F + j]; a = a > 0 ? a : 0; err = fmax(err, fabs(a - hG[(size_t)tile * I + s * F + j])); }
  }
  printf("max |g - host fp64| over 4 tiles: %.3g\n", err);
  return 0;
}
